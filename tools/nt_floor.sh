# NT GEMM floor table of the replayed step (tools/nt_floor.py): bash tools/nt_floor.sh [bench.py args] > profiles/rNN_nt_floor.txt   (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ntf
python tools/nt_floor.py --metas gpurun_out/nt_metas.json -- --steps 2 --warmup 2 --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/ntf_metas.err || { tail -5 gpurun_out/ntf_metas.err; exit 1; }
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ntf -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-trace "$@" > gpurun_out/ntf.log 2>&1
python tools/nt_floor.py --table gpurun_out/nt_metas.json gpurun_out/ntf
rm -rf gpurun_out/ntf
