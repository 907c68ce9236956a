# device ISA + resource usage of one source of the library:  bash tools/diag/isa_of.sh attn_fwd [extra -D flags] -> /tmp/isa/<name>.s
# (hipcc -S of the device side only; CPU container, no GPU needed)
set -e
name=$1; shift
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -ffast-math -fno-finite-math-only -fno-slp-vectorize -Wno-unused-result \
  --cuda-device-only -S "$@" /root/repo/svit_amd/csrc/$name.hip -o /tmp/isa/$name.s
grep -E "^\s+\.(vgpr_count|sgpr_count|agpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size)|^\s+\.name:" /tmp/isa/$name.s | paste - - - - - - - - | sed 's/\s\+/ /g'
