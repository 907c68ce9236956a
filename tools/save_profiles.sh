# copy the outputs of tools/collect_profiles.sh from gpurun_out/ into profiles/ under a version tag:
#   bash tools/save_profiles.sh r02_v2
set -e
V=$1
cp gpurun_out/bN.log profiles/${V}_bench.json
cp gpurun_out/bNf.log profiles/${V}_bench_frames_pass.json
cp gpurun_out/rNN_bench_kernel_stats.csv profiles/${V}_bench_kernel_stats.csv
cp gpurun_out/rNN_pmc_traffic.txt profiles/${V}_pmc_traffic.txt
cp gpurun_out/rNN_device_busy.txt profiles/${V}_device_busy.txt
cp gpurun_out/rNN_backward_by_block.txt profiles/${V}_backward_by_block.txt
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
