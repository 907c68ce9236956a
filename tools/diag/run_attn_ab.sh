# A/B of a diagnostic variant of the attention forward against the product library (GPU box):
#   bash tools/diag/run_attn_ab.sh <variant tag> [label]
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  python tools/diag/attn_fwd_ablate.py product
  SVIT_HIP_LIB=tools/diag/libsvit_diag_$1.so python tools/diag/attn_fwd_ablate.py "${2:-$1}" 2>/dev/null
done
