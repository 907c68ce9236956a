# Collects the artefacts kept under profiles/ (run on the GPU box: GIT_HEAD=<sha> bash tools/collect_profiles.sh;
# outputs in gpurun_out/, copied into profiles/ by tools/save_profiles.sh <tag>).  The PMC passes run first, so the
# bench line that follows quotes the traffic measured on THIS build (roofline.traffic / traffic_head).
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/profN gpurun_out/pmcNf gpurun_out/pmcNw
echo "[collect] PMC pass 1/2 (FETCH_SIZE)"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcNf -- python bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > gpurun_out/pmcNf.log 2>&1
echo "[collect] PMC pass 2/2 (WRITE_SIZE)"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcNw -- python bench.py --steps 1 --warmup 1 --eager --no-cpu-baseline --no-kernel-trace > gpurun_out/pmcNw.log 2>&1
F=$(ls gpurun_out/pmcNf/*/*counter_collection.csv | head -1); W=$(ls gpurun_out/pmcNw/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W > gpurun_out/rNN_pmc_traffic.txt
cp profiles/pmc_traffic.json gpurun_out/pmc_traffic_prev.json 2>/dev/null || true
python tools/pmc_to_json.py gpurun_out/rNN_pmc_traffic.txt "${GIT_HEAD:-unknown}" > /dev/null && cp profiles/pmc_traffic.json gpurun_out/pmc_traffic.json
echo "[collect] bench lines"
python bench.py --steps 30 --warmup 5 > gpurun_out/bN.log 2>gpurun_out/bN.err
python bench.py --steps 20 --warmup 5 --frames-pass --no-cpu-baseline --no-kernel-trace > gpurun_out/bNf.log 2>gpurun_out/bNf.err
echo "[collect] kernel trace"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/profN -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-trace > gpurun_out/profN.log 2>&1
K=$(ls gpurun_out/profN/*/*kernel_trace.csv | head -1)
python tools/trace_gaps.py $K 3000 > gpurun_out/rNN_device_busy.txt
python tools/trace_gaps.py $K --blocks > gpurun_out/rNN_backward_by_block.txt
cp $(ls gpurun_out/profN/*/*kernel_stats.csv | head -1) gpurun_out/rNN_bench_kernel_stats.csv
# keep the merge small
rm -rf gpurun_out/profN gpurun_out/pmcNf gpurun_out/pmcNw
tail -1 gpurun_out/bN.log | cut -c1-300; tail -1 gpurun_out/bNf.log | cut -c1-200; head -3 gpurun_out/rNN_device_busy.txt
