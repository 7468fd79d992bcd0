#!/bin/bash
# compile one csrc/*.hip for gfx950, keep the ISA in /tmp, print register / spill counts per kernel; for wide.hip (which owns the
# accumulator file by name) also the number of compiler-made v_accvgpr_* instructions outside its inline-asm blocks: must be 0
f=$1
cd /root/repo/scenerf_amd/csrc || exit 1
extra=""
[ "$f" = "wide" ] && extra="-mllvm -amdgpu-spill-vgpr-to-agpr=0"
rm -f /tmp/$f.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -munsafe-fp-atomics $extra -c $f.hip -o /tmp/$f.o -save-temps=obj 2>&1 | grep -E "error|warning" | head -20; [ -s /tmp/$f.o ] || { echo "COMPILE FAILED"; exit 1; }
grep -E "^\s+\.(name|vgpr_count|vgpr_spill_count|sgpr_count|agpr_count|group_segment_fixed_size|private_segment_fixed_size):" /tmp/$f-hip-amdgcn-amd-amdhsa-gfx950.s | paste - - - - - - - | sed 's/  */ /g'
python3 - "$f" <<'PY'
import sys
s = open("/tmp/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % sys.argv[1]).read()
inasm, out = False, 0
for l in s.split("\n"):
    if "#ASMSTART" in l: inasm = True
    elif "#ASMEND" in l: inasm = False
    elif "v_accvgpr" in l and not inasm: out += 1
print("compiler-made v_accvgpr_* outside inline asm:", out)
PY
