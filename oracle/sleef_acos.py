"""ORACLE, TEST INFRASTRUCTURE ONLY: torch's own SLEEF routines, callable on tensors.

Why this exists.  The reference's sphere pixel is ``round()`` of an angle that went through ``torch.acos`` / ``torch.atan2``
(spherical_mapping.py:99-115).  On the CPU, ``torch.atan2`` (float32) is SLEEF's ``Sleef_atan2f{8,16}_u10`` -- the same bits under AVX2
and AVX-512 -- but ``torch.acos`` is Intel MKL's VML ``vmsAcos(VML_HA)`` wherever torch is built with MKL (ATen/cpu/vml.h), a
closed-source routine whose result depends on the instruction set MKL dispatches to: on this project's build container the AVX-512 and
AVX2 kernels disagree on 207 of 4,194,304 inputs, the SSE4.2 kernel on 11,570 (``tools/sleef_check/acos_isa_probe.py``,
``profiles/r05_acos_isa_probe.txt``), and a torch built without MKL (aarch64 wheels) calls SLEEF's ``Sleef_acosf*_u10`` instead.  The
reference therefore has no single answer in the last bit of acos, and a sample within an ulp of a .5 boundary lands on either texel
depending on the host.  The rule this project pins: **acos = SLEEF u10**, the open routine torch itself ships and uses
for the sibling op, identical across its AVX2 / AVX-512 builds.  ``OracleConfig.index_rule = "pinned"`` selects it (together with the k-ordered fma
chains for the path's 3x3 / 4x4 products: MKL's sgemm sums them in another order on other CPUs, see OracleConfig); the default (``"torch"``)
stays the reference's calls, whatever the host makes of them.

The functions below call the routine inside torch's own ``libtorch_cpu.so`` (exported symbols ``Sleef_acosf8_u10avx2`` /
``Sleef_atan2f8_u10avx2``) through ``oracle/sleef_shim.c``: the oracle does not restate SLEEF, it runs it.  Needs gcc and an AVX2 host.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
LIB = os.path.join(OUT, "libsleef_shim.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(HERE, "sleef_shim.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        os.makedirs(OUT, exist_ok=True)
        tl = os.path.join(os.path.dirname(torch.__file__), "lib")
        subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-shared", "-fPIC", "-o", LIB, src, "-L" + tl, "-ltorch_cpu",
                               "-Wl,-rpath," + tl, "-lm"])
    return LIB


def _load():
    global _lib
    if _lib is None:
        try:
            _lib = C.CDLL(build())
        except OSError:                      # a library built against another torch install travelled here: rebuild
            _lib = C.CDLL(build(force=True))
        _lib.oracle_sleef_acosf.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _lib.oracle_sleef_atan2f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib.oracle_matvec_fma.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]
    return _lib


def acos(x: torch.Tensor) -> torch.Tensor:
    """SLEEF acosf u10 (torch's build) of a float32 CPU tensor."""
    assert x.dtype == torch.float32 and not x.is_cuda
    xc = x.detach().contiguous()
    y = torch.empty_like(xc)
    if xc.numel():
        _load().oracle_sleef_acosf(xc.data_ptr(), y.data_ptr(), xc.numel())
    return y.reshape(x.shape)


def atan2(p: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """SLEEF atan2f u10 (torch's build): what torch.atan2 itself returns on an AVX2 / AVX-512 host."""
    assert p.dtype == torch.float32 and q.dtype == torch.float32 and p.shape == q.shape
    pc, qc = p.detach().contiguous(), q.detach().contiguous()
    y = torch.empty_like(pc)
    if pc.numel():
        _load().oracle_sleef_atan2f(pc.data_ptr(), qc.data_ptr(), y.data_ptr(), pc.numel())
    return y.reshape(p.shape)


def matvec_fma(A: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    """(A @ X.T).T for A (rows, k), X (M, k), float32 on the CPU, every element a k-ordered fma chain (sleef_shim.c: oracle_matvec_fma)."""
    assert A.dtype == torch.float32 and X.dtype == torch.float32 and A.dim() == 2 and X.dim() == 2 and A.shape[1] == X.shape[1]
    Ac, Xc = A.detach().contiguous(), X.detach().contiguous()
    out = torch.empty((Xc.shape[0], Ac.shape[0]), dtype=torch.float32)
    if Xc.shape[0]:
        _load().oracle_matvec_fma(Ac.data_ptr(), Ac.shape[0], Ac.shape[1], Xc.data_ptr(), Xc.shape[0], out.data_ptr())
    return out


if __name__ == "__main__":
    print(build(force=True))
