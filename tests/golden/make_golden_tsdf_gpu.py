"""Golden volumes for the TSDF "gpu" update rule from the REFERENCE's own kernel: the pycuda SourceModule string of
scenerf/data/utils/fusion.py:72-145, extracted from the mounted reference and compiled verbatim with hipcc for gfx950 by
oracle/build_ref.py (build container) -- once with every operation rounded on its own (-ffp-contract=off: THE PIN, keys ``*_v08`` /
``*_v07``) and once under the compiler's default contraction (keys ``*_contract``, a statistic) -- then run HERE on an MI355X:

    python oracle/build_ref.py                                      # build container (needs /root/reference)
    gpurun -- python tests/golden/make_golden_tsdf_gpu.py            # GPU box: writes gpurun_out/tsdf_gpu_semantics.npz
    cp gpurun_out/tsdf_gpu_semantics.npz tests/golden/

The launch geometry follows fusion.py:148-156 (block = MAX_THREADS_PER_BLOCK, cube-ish grid, n_gpu_loops); the scene is
tests/tsdf_scene.py (seed 7, three frames), at the two voxel sizes tests/test_tsdf.py uses (z extent 44 and 50)."""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def run_reference_kernel(sc, lib):
    import tsdf_oracle as orc
    dim, origin, tsdf, weight, color = orc.new_volume(sc["vol_bnds"], sc["voxel_size"])      # fusion.py:33-56
    n = int(np.prod(dim))
    fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    vol = C.c_void_p()
    assert lib.tsdf_ref_create(fp(tsdf), fp(weight), fp(color), C.c_longlong(n), C.byref(vol)) == 0, lib.tsdf_ref_last_error()
    # fusion.py:148-156
    threads = 1024
    n_blocks = int(np.ceil(float(n) / float(threads)))
    gx = min(2 ** 31 - 1, int(np.floor(np.cbrt(n_blocks))))
    gy = min(65535, int(np.floor(np.sqrt(n_blocks / gx))))
    gz = min(65535, int(np.ceil(float(n_blocks) / float(gx * gy))))
    loops = int(np.ceil(float(n) / float(gx * gy * gz * threads)))
    vol_dim = np.asarray(dim).astype(np.float32)
    for fr in sc["frames"]:
        col = orc.fold_color(fr["color"]).reshape(-1).astype(np.float32)                        # fusion.py:218-220
        dep = fr["depth"].reshape(-1).astype(np.float32)
        K = np.asarray(sc["cam_intr"]).reshape(-1).astype(np.float32)
        P = np.asarray(fr["pose"]).reshape(-1).astype(np.float32)
        im_h, im_w = fr["depth"].shape
        rc = lib.tsdf_ref_integrate(vol, fp(vol_dim), fp(origin), fp(K), fp(P), C.c_float(sc["voxel_size"]), im_h, im_w,
                                    C.c_float(sc["trunc_margin"]), C.c_float(1.0), fp(col), fp(dep), threads, gx, gy, gz, loops)
        assert rc == 0, lib.tsdf_ref_last_error()
    assert lib.tsdf_ref_read(vol, fp(tsdf), fp(weight), fp(color)) == 0
    lib.tsdf_ref_destroy(vol)
    return tsdf, weight, color


if __name__ == "__main__":
    import tsdf_scene
    out = {}
    # the pin (every operation rounded on its own) and the same text under the compiler's default contraction (oracle/build_ref.py)
    for suffix, name in (("", "libtsdf_ref.so"), ("_contract", "libtsdf_ref_contract.so")):
        lib = C.CDLL(os.path.join(ROOT, "oracle", "_ref", name))
        lib.tsdf_ref_last_error.restype = C.c_char_p
        for tag, vs in (("v08", 0.08), ("v07", 0.07)):
            sc = tsdf_scene.make(seed=7)
            sc["voxel_size"] = vs
            t, w, c = run_reference_kernel(sc, lib)
            out.update({"tsdf_" + tag + suffix: t, "weight_" + tag + suffix: w, "color_" + tag + suffix: c})
            print("%s%s: dims %s, %d voxels observed, %d at 255" % (tag, suffix, t.shape, int((w > 0).sum()), int((t == 255).sum())))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    dst = os.path.join(ROOT, "gpurun_out", "tsdf_gpu_semantics.npz")
    np.savez_compressed(dst, seed=7, **out)
    print("wrote", dst, os.path.getsize(dst))
