cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention or pool_ln" > gpurun_out/r06_t4_tests.log 2>&1 || { tail -40 gpurun_out/r06_t4_tests.log; exit 1; }
tail -2 gpurun_out/r06_t4_tests.log
run() { python tools/diag/engine_attr_ab.py "$@" -- --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%-20s %.3f ms' % (' '.join(d['engine_attrs']), d['ms_per_step']))"; }
for rep in 1 2 3; do run --attr fused_qln=0; run --attr fused_qln=1; done
