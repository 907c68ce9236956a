cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/dkv_tr
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dkv_tr -- python tools/attn_dkv_vs_nq.py > gpurun_out/dkv_tr.log 2>&1
cat gpurun_out/dkv_tr.log | tail -16
