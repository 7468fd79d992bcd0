"""Is torch-CPU's float32 acos / atan2 one function?  ("if torch-CPU turns out not to be self-consistent across
ISAs, commit the probe that shows it".)

Runs the same 4,194,304 inputs through torch.acos / torch.atan2 in child processes that differ only in the instruction set MKL
(MKL_ENABLE_INSTRUCTIONS) and ATen (ATEN_CPU_CAPABILITY) may dispatch to, and compares the result bits; it also compares with MKL's
vmsAcos called directly, with torch's own SLEEF build (oracle/sleef_acos.py) and with the correctly rounded value.

    python tools/sleef_check/acos_isa_probe.py > profiles/r05_acos_isa_probe.txt
"""
import ctypes as C, hashlib, os, subprocess, sys, tempfile
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = 1 << 22


def inputs():
    x = torch.rand(N, generator=torch.Generator().manual_seed(0)) * 2 - 1
    a = torch.randn(N, generator=torch.Generator().manual_seed(1))
    b = torch.randn(N, generator=torch.Generator().manual_seed(2))
    return x, a, b


if len(sys.argv) > 1 and sys.argv[1] == "child":
    x, a, b = inputs()
    torch.save((torch.acos(x), torch.atan2(a, b), torch.backends.cpu.get_cpu_capability()), sys.argv[2])
    sys.exit(0)

ne = lambda p, q: int((p.view(torch.int32) != q.view(torch.int32)).sum())
runs = {}
with tempfile.TemporaryDirectory() as td:
    for name, env in (("host default", {}), ("MKL_ENABLE_INSTRUCTIONS=AVX2", {"MKL_ENABLE_INSTRUCTIONS": "AVX2"}),
                      ("MKL_ENABLE_INSTRUCTIONS=SSE4_2", {"MKL_ENABLE_INSTRUCTIONS": "SSE4_2"}),
                      ("ATEN_CPU_CAPABILITY=avx2", {"ATEN_CPU_CAPABILITY": "avx2"}),
                      ("ATEN_CPU_CAPABILITY=default", {"ATEN_CPU_CAPABILITY": "default"})):
        out = os.path.join(td, "o.pt")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", out], env=dict(os.environ, **env))
        runs[name] = torch.load(out)
x, a, b = inputs()
base = runs["host default"]
print("torch", torch.__version__, "| ATen capability on this host:", base[2], "| inputs:", N)
print("%-34s %-10s %22s %22s" % ("run", "ATen cap", "acos bits != default", "atan2 bits != default"))
for name, (ac, at, cap) in runs.items():
    print("%-34s %-10s %22d %22d" % (name, cap, ne(ac, base[0]), ne(at, base[1])))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sleef_acos
sl_acos, sl_atan2 = sleef_acos.acos(x), sleef_acos.atan2(a, b)
cr = torch.acos(x.double()).float()
print()
print("torch.acos (host default) vs torch's own SLEEF Sleef_acosf8_u10avx2 : %d differ" % ne(base[0], sl_acos))
print("torch.acos (host default) vs correctly rounded (float64 acos -> f32) : %d differ" % ne(base[0], cr))
print("SLEEF acosf u10           vs correctly rounded                       : %d differ" % ne(sl_acos, cr))
print("torch.atan2 (host default) vs torch's own SLEEF Sleef_atan2f8_u10avx2: %d differ" % ne(base[1], sl_atan2))
try:
    L = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libtorch_cpu.so"))
    L.vmsAcos.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64]
    y = torch.empty(N)
    L.vmsAcos(N, x.data_ptr(), y.data_ptr(), 0x2 | 0x00140000 | 0x100)      # VML_HA | VML_FTZDAZ_OFF | VML_ERRMODE_IGNORE (ATen/cpu/vml.h)
    print("torch.acos (host default) vs MKL vmsAcos(VML_HA) called directly     : %d differ" % ne(base[0], y))
except (OSError, AttributeError) as e:
    print("MKL vmsAcos not exported by this torch build:", e)
print()
print("reading: torch.acos on this host IS MKL's vmsAcos(HA); its bits change with the instruction set MKL dispatches to; torch.atan2 is")
print("SLEEF wherever ATen runs a vector kernel (ATEN_CPU_CAPABILITY=default is the scalar std::atan2 path).  SLEEF's acosf is the same")
print("under AVX2 and AVX-512 (tests/test_sphere_exact.py).  The sphere index rule pinned by this project is SLEEF u10 for both.")
