"""Annotated disassembly of a SLEEF routine inside torch's libtorch_cpu.so (SLEEF is Boost-licensed, vendored by torch as
third_party/sleef): constants referenced through %rip are resolved to their float values, so the operation sequence of
scenerf_amd/csrc/sphere_exact.h can be read against it instruction by instruction.

    python tools/sleef_check/sleef_dis.py Sleef_acosf16_u10
    python tools/sleef_check/sleef_dis.py Sleef_atan2f16_u10
"""
import os, re, struct, subprocess, sys
import torch

L = os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so")
sym = sys.argv[1]
if not sym.startswith("Sleef_"):
    sys.exit("only SLEEF symbols (open source) are meant to be read with this tool")
nm = subprocess.check_output(["nm", "-D", "-S", "--defined-only", L]).decode()
rows = sorted((int(a, 16), n) for a, *_, n in (ln.split() for ln in nm.splitlines() if len(ln.split()) >= 3))
addr = [a for a, n in rows if n == sym][0]
stop = min(a for a, _ in rows if a > addr)
segs = []
for ln in subprocess.check_output(["readelf", "-lW", L]).decode().splitlines():
    m = re.match(r"\s*LOAD\s+(0x[0-9a-f]+)\s+(0x[0-9a-f]+)\s+(0x[0-9a-f]+)\s+(0x[0-9a-f]+)", ln)
    if m:
        segs.append(tuple(int(x, 16) for x in m.groups()))
f = open(L, "rb")


def rd(va):
    for off, vaddr, _, fsz in segs:
        if vaddr <= va < vaddr + fsz:
            f.seek(off + va - vaddr)
            b = f.read(4)
            return struct.unpack("<f", b)[0], struct.unpack("<I", b)[0]
    return None, None


d = subprocess.check_output(["objdump", "-d", "--no-show-raw-insn", "--start-address=%#x" % addr, "--stop-address=%#x" % stop, L]).decode()
for ln in d.splitlines():
    m = re.search(r"#\s*([0-9a-f]+)\s*<", ln)
    ln = re.sub(r"<_ZTS[^>]*>|<[^>]*\+0x[0-9a-f]+>", "", ln)
    if m:
        v, u = rd(int(m.group(1), 16))
        ln += "   ; = %r (0x%08x)" % (v, u)
    print(ln)
