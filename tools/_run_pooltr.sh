cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pool_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pool_tr -- python tools/bench_kernels.py pooltiled > gpurun_out/pool_tr.log 2>&1
tail -12 gpurun_out/pool_tr.log
