# round 6: the 256x192 / 8-wave grouped TN tile -- parity, then in-step A/B against the product's 128x192 (alternating, one box)
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm_tn" > gpurun_out/r06_tn_tests.log 2>&1 || { tail -30 gpurun_out/r06_tn_tests.log; exit 1; }
tail -2 gpurun_out/r06_tn_tests.log
run() { python tools/bench_knobs.py "$@" -- --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-40s %.3f ms' % (' '.join(d['knobs']), d['ms_per_step']))"; }
for rep in 1 2; do
  run --set tn_tile=2
  run --set tn_tile=4
  run --set tn_tile=4 --set tn=70,75
  run --set tn_tile=4 --set tn=100,75
  run --set tn_tile=4 --set tn=85,120
done
