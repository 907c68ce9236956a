cd $GRAFT_REPO_ROOT
for m in 0 17 255; do python tools/attn_ap_stamps.py 1633 1633 160 4 $m 2>&1 | grep -v Warn\|warn\|amdgpu.ids; done > gpurun_out/r06_ap_stamps.txt
python tools/attn_ap_stamps.py 1633 1633 160 4 0 -DSVIT_ATTN_AP_PRIO=0 2>&1 | grep -v Warn\|warn\|amdgpu.ids >> gpurun_out/r06_ap_stamps.txt
cat gpurun_out/r06_ap_stamps.txt
