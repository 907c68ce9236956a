# Round-3 baseline on one box: per-shape attention / pooling micro-benchmarks at C2, and the
# C4 (32x224^2) and C5 (16x312^2 3-crop eval) single-GPU measurements with kernel breakdown.
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03base; mkdir -p $O
python tools/bench_kernels.py attn > $O/attn.txt 2>&1
python tools/bench_kernels.py attnfwd > $O/attnfwd.txt 2>&1
python tools/bench_kernels.py pooltiled > $O/pool.txt 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2.json 2>$O/c2.err
python bench.py --frames 32 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/c4_b4.json 2>$O/c4_b4.err
python bench.py --frames 32 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > $O/c4_b8.json 2>$O/c4_b8.err
python tools/bench_eval.py --crop 312 --videos 2 > $O/c5_eval.json 2>$O/c5_eval.err
python tools/bench_eval.py --crop 224 --videos 4 > $O/c2_eval.json 2>$O/c2_eval.err
tail -n 12 $O/attn.txt; cut -c1-400 $O/c2.json; cut -c1-300 $O/c4_b4.json; cut -c1-300 $O/c4_b8.json; cat $O/c5_eval.json
