# Round-4 measurement set on ONE box (GIT_HEAD=<sha> bash tools/r04_collect.sh): the headline bench with PMC traffic + rocprof stats
# (tools/collect_profiles.sh), attention per shape for C2 and C4, the C4 / C5 single-GPU configurations.
set -e
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh
O=gpurun_out
python tools/bench_kernels.py attn c2 > $O/r04_attn_c2.txt 2>/dev/null
python tools/bench_kernels.py attn c4 > $O/r04_attn_c4.txt 2>/dev/null
python bench.py --frames 32 --batch 4 --steps 10 --warmup 3 --no-cpu-baseline > $O/r04_c4_b4.json 2>$O/r04_c4_b4.err
python tools/bench_eval.py --crop 312 --videos 2 > $O/r04_c5_eval.json 2>$O/r04_c5_eval.err
tail -n 9 $O/r04_attn_c2.txt; tail -n 9 $O/r04_attn_c4.txt; cut -c1-200 $O/r04_c4_b4.json; cat $O/r04_c5_eval.json
