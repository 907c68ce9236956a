# round 6, call 2: the anti-phase forward kernel -- parity tests, per-shape timing, in-step A/B
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -k "anti_phase or attention_fwd" > gpurun_out/r06_ap_tests.log 2>&1 || { tail -30 gpurun_out/r06_ap_tests.log; exit 1; }
tail -3 gpurun_out/r06_ap_tests.log
timeout -k 10 200 python tools/bench_kernels.py attnap c2 > gpurun_out/r06_ap_per_shape.txt 2>&1 && timeout -k 10 200 python tools/bench_kernels.py attnap c4 >> gpurun_out/r06_ap_per_shape.txt 2>&1
grep blk gpurun_out/r06_ap_per_shape.txt
for rep in 1 2; do
for k in 0 12; do
python tools/bench_knobs.py --set attn:4=$k -- --steps 30 --warmup 5 --no-cpu-baseline --no-kernel-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('ap=$k %.3f ms' % d['ms_per_step'])"
done; done
