"""Build libscenerf_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m scenerf_amd.build [--force]

The library is a plain C-ABI shared object (include/scenerf_hip.h); no torch headers are involved.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# development knobs (kernel experiments, tools/variants.sh): extra -D flags build a separately named library next to the default one
_TAG = os.environ.get("SRF_LIB_TAG", "")
LIB = os.path.join(CSRC, "libscenerf_hip%s.so" % ("_" + _TAG if _TAG else ""))
SOURCES = ["runtime.hip", "rays.hip", "gemm.hip", "wgrad.hip", "fused.hip", "wide.hip", "dfeat.hip", "mlp.hip", "loss.hip", "tsdf.hip", "sphere.hip", "optim.hip"]
# wide.hip owns the whole accumulator file (a[0:255] by name in inline-asm MFMAs): the compiler must not park spilled VGPRs there
# (a spill then shows up as scratch usage, which tools/asmcheck.sh and the build's resource check refuse)
EXTRA = {"wide.hip": ["-mllvm", "-amdgpu-spill-vgpr-to-agpr=0"]}
HEADERS = ["common.h", "gemm.h", "fused.h", "sphere_exact.h", os.path.join("..", "..", "include", "scenerf_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics",
         "-Wall", "-Wno-unused-function"] + os.environ.get("SRF_EXTRA_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


STAMP = LIB + ".stamp"


def _digest() -> str:
    """Content hash of sources + flags (mtimes are meaningless after the tree is copied to the GPU box)."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + repr(sorted(EXTRA.items()))).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != _digest()


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", (("_" + _TAG) if _TAG else "") + ".o"))
        cmd = [hipcc] + FLAGS + EXTRA.get(src, []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode != 0:
            failed = True
            print("FAILED: %s\n%s" % (src, out), file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP, "w") as fh:
        fh.write(_digest())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
