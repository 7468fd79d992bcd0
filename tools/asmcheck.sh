#!/bin/bash
# compile one csrc/*.hip for gfx950, keep the ISA in /tmp, print register / spill counts per kernel
f=$1
cd /root/repo/scenerf_amd/csrc || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics -c $f.hip -o /tmp/$f.o -save-temps=obj 2>&1 | grep -E "error|warning" | head -20
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_count|agpr_count|group_segment_fixed_size):" /tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - - | sed 's/  */ /g'
