# in-step A/B of diagnostic library variants (tools/diag/build_variant.py) against the product library, alternating on one box:
#   bash tools/diag/lib_ab.sh <tag> [<tag> ...] [-- bench.py args]      (GPU box)
cd $GRAFT_REPO_ROOT
TAGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do TAGS+=("$1"); shift; done; [ "$1" = "--" ] && shift
for rep in 1 2; do
  for t in product "${TAGS[@]}"; do
    if [ $t = product ]; then unset SVIT_HIP_LIB; else export SVIT_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/libsvit_diag_$t.so; fi
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-10s %.3f ms' % ('$t', d['ms_per_step']))"
  done
done
