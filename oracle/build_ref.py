"""Build oracle/_ref/ from the mounted reference (TEST INFRASTRUCTURE ONLY; build container only: /root/reference is not on the GPU box).

    python oracle/build_ref.py

* ``libtsdf_ref.so``: the reference's pycuda TSDF kernel (scenerf/data/utils/fusion.py:72-145) compiled verbatim with hipcc for gfx950
  behind oracle/tsdf_ref_host.hip.  The kernel text is read out of the reference file at build time and written to
  ``oracle/_ref/tsdf_ref_kernel.inc``; nothing of it is committed (``oracle/_ref/`` is git-ignored, but travels to the GPU box with
  gpurun like every other built artefact).  Two builds of the same text:
    - ``libtsdf_ref.so``  with ``-ffp-contract=off``: every operation of the kernel source rounded on its own (IEEE fp32, correctly
      rounded division: hipcc's default, as nvcc's -prec-div=true).  This is the only compiler-independent reading of the source and
      THE PIN: oracle/tsdf_oracle.py::integrate_gpu_semantics and scenerf_amd/csrc/tsdf.hip must reproduce its volumes bit for bit.
    - ``libtsdf_ref_contract.so`` with the compiler's default contraction (a*b+c -> fma where the compiler chooses to; nvcc's default
      --fmad=true does the same under pycuda, with ITS OWN choice of which products to fuse -- hipcc e.g. fuses two of the three
      rotation rows completely and the third only partly).  Kept as a statistic: how many voxels a different fusion choice moves.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/scenerf/data/utils/fusion.py"
OUT = os.path.join(HERE, "_ref")


def extract_kernel() -> str:
    src = open(REF).read()
    m = re.search(r'SourceModule\("""(.*?)"""\)', src, re.S)
    if not m:
        raise RuntimeError("no SourceModule string in %s" % REF)
    text = m.group(1)
    if "__global__ void integrate(" not in text:
        raise RuntimeError("unexpected kernel text")
    return text


def build(verbose: bool = True) -> str:
    if not os.path.exists(REF):
        raise RuntimeError("the reference is not mounted here")
    os.makedirs(OUT, exist_ok=True)
    inc = os.path.join(OUT, "tsdf_ref_kernel.inc")
    with open(inc, "w") as f:
        f.write("// extracted from %s (SourceModule string) by oracle/build_ref.py -- DO NOT COMMIT\n" % REF)
        f.write(extract_kernel() + "\n")
    lib = os.path.join(OUT, "libtsdf_ref.so")
    for out, extra in ((lib, ["-ffp-contract=off"]), (os.path.join(OUT, "libtsdf_ref_contract.so"), [])):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared"] + extra + ["-I", HERE, os.path.join(HERE, "tsdf_ref_host.hip"),
                                                                                                     "-o", out]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build())
