# per-launch fixed cost of svit_attn_fwd by ablation (GPU box): bash tools/diag/run_attn_ablate.sh
cd $GRAFT_REPO_ROOT
python tools/diag/attn_fwd_ablate.py product
for m in 1 2 4 8 15; do
  case $m in 1) L="no-Q-loads";; 2) L="no-residual-loads";; 4) L="no-ctx-stores";; 8) L="no-KV-DMA";; 15) L="none-of-the-four";; esac
  SVIT_HIP_LIB=tools/diag/libsvit_diag_attnabl$m.so python tools/diag/attn_fwd_ablate.py "$L" 2>/dev/null
done
python tools/diag/attn_fwd_ablate.py product-again
