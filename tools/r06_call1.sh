# round 6, call 1: baseline on this box + static-priority variants of the attention kernels (in-step and per shape)
cd $GRAFT_REPO_ROOT
echo "[call1] in-step A/B"
bash tools/diag/lib_ab.sh prio1 prio2 prio3 > gpurun_out/r06_prio_instep.txt 2>&1
cat gpurun_out/r06_prio_instep.txt
echo "[call1] per shape"
for t in product prio1 prio2 prio3; do
  if [ $t = product ]; then unset SVIT_HIP_LIB; else export SVIT_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/libsvit_diag_$t.so; fi
  echo "== $t" >> gpurun_out/r06_prio_per_shape.txt
  python tools/bench_kernels.py attn c2 >> gpurun_out/r06_prio_per_shape.txt 2>&1
done
unset SVIT_HIP_LIB
export SVIT_HIP_LIB=$GRAFT_REPO_ROOT/tools/diag/libsvit_diag_prio1.so
echo "== prio1 c4" >> gpurun_out/r06_prio_per_shape.txt
python tools/bench_kernels.py attn c4 >> gpurun_out/r06_prio_per_shape.txt 2>&1
unset SVIT_HIP_LIB
echo "== product c4" >> gpurun_out/r06_prio_per_shape.txt
python tools/bench_kernels.py attn c4 >> gpurun_out/r06_prio_per_shape.txt 2>&1
grep "step totals\|==" gpurun_out/r06_prio_per_shape.txt
