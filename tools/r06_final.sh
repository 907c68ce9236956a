# end-of-round collection (GPU box): GIT_HEAD=<sha> bash tools/r06_final.sh   -> gpurun_out/ (copied to profiles/ by hand afterwards)
cd $GRAFT_REPO_ROOT
GIT_HEAD=${GIT_HEAD:-unknown} bash tools/collect_profiles.sh > gpurun_out/collect.log 2>&1 || { tail -20 gpurun_out/collect.log; exit 1; }
tail -4 gpurun_out/collect.log
echo "[final] step breakdown"
bash tools/step_breakdown.sh > gpurun_out/r06_step_breakdown.txt 2>&1
head -10 gpurun_out/r06_step_breakdown.txt
echo "[final] C4 32x224^2 B=4 with the kernel trace (roofline_attn_* on record)"
python bench.py --frames 32 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_c4.json 2> gpurun_out/r06_c4.err
tail -1 gpurun_out/r06_c4.json | cut -c1-200
echo "[final] C1 8x224^2 B=8"
python bench.py --frames 8 --batch 8 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-trace > gpurun_out/r06_c1.json 2> gpurun_out/r06_c1.err
tail -1 gpurun_out/r06_c1.json | cut -c1-160
echo "[final] C5 3-crop eval 312^2 / 224^2"
python tools/bench_eval.py --crop 312 --videos 2 > gpurun_out/r06_c5_eval.json 2> gpurun_out/r06_c5.err; tail -1 gpurun_out/r06_c5_eval.json | cut -c1-200
python tools/bench_eval.py --crop 224 --videos 4 > gpurun_out/r06_eval224.json 2> gpurun_out/r06_e224.err; tail -1 gpurun_out/r06_eval224.json | cut -c1-200
echo "[final] throughput vs batch"
for b in 1 2 4 8 12 16; do python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-trace 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B=%-3d %.3f ms/step  %.1f clips/s' % ($b, d['ms_per_step'], d['value']))"; done | tee gpurun_out/r06_throughput_vs_batch.txt
echo "[final] attention per shape"
python tools/bench_kernels.py attn c2 2>&1 | grep "blk\|totals\|==" > gpurun_out/r06_attn_per_shape.txt; python tools/bench_kernels.py attn c4 2>&1 | grep "blk\|totals\|==" >> gpurun_out/r06_attn_per_shape.txt
grep totals gpurun_out/r06_attn_per_shape.txt
