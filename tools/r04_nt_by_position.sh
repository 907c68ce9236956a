# Position-matched NT GEMM table of one replayed step: A = the 128-row tiles only (SVIT_NT_ONE_ROUND=0), B = this build's heuristic
# (160x256 / 192x192 one-round tiles where the grid fits).   GPU box: bash tools/r04_nt_by_position.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ntA $R/gpurun_out/ntB
SVIT_NT_ONE_ROUND=0 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ntA -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/ntA.log 2>&1
SVIT_NT_ONE_ROUND=1 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ntB -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-trace > $R/gpurun_out/ntB.log 2>&1
cd $R
python3 tools/nt_by_position.py gpurun_out/ntA gpurun_out/ntB > gpurun_out/r04_nt_by_position.txt 2>&1
rm -rf gpurun_out/ntA gpurun_out/ntB
cat gpurun_out/r04_nt_by_position.txt
