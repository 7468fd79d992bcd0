"""Golden vectors for the TSDF fusion from the reference itself: scenerf/data/utils/fusion.py imported with stub
`numba` (njit = identity, prange = range) and `skimage` packages -- neither is installed here, and neither touches the arithmetic --
and pycuda absent, i.e. the reference's own CPU path (FUSION_GPU_MODE = 0).  Run in the build container only (needs /root/reference):
    python tests/golden/make_golden_tsdf.py
The scene is synthetic and regenerated from the seed by tests/tsdf_scene.py; only the resulting volumes are stored."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = "/root/reference"


def _install_stubs():
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if (len(a) == 1 and callable(a[0])) else (lambda f: f))
    nb.prange = range
    sys.modules["numba"] = nb
    sk = types.ModuleType("skimage")
    sk.measure = types.ModuleType("skimage.measure")
    sys.modules["skimage"] = sk
    sys.modules["skimage.measure"] = sk.measure
    sys.path.insert(0, REF)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    _install_stubs()
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_fusion", os.path.join(REF, "scenerf", "data", "utils", "fusion.py"))
    fusion = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fusion)
    assert fusion.FUSION_GPU_MODE == 0
    import tsdf_scene
    sc = tsdf_scene.make(seed=7)
    vol = fusion.TSDFVolume(sc["vol_bnds"].copy(), voxel_size=sc["voxel_size"], trunc_margin=sc["trunc_margin"], use_gpu=False)
    for fr in sc["frames"]:
        vol.integrate(fr["color"], fr["depth"], sc["cam_intr"], fr["pose"], obs_weight=1.0)
    tsdf, color = vol.get_volume()
    out = os.path.join(HERE, "tsdf_cpu_semantics.npz")
    np.savez_compressed(out, tsdf=tsdf.astype(np.float32), color=color.astype(np.float32), weight=vol._weight_vol_cpu.astype(np.float32),
                        vol_dim=np.asarray(vol._vol_dim), seed=7)
    print("wrote %s (%.1f KB): dims %s, %d voxels observed, %d at the initial 255" % (
        out, os.path.getsize(out) / 1024, tuple(vol._vol_dim), int((vol._weight_vol_cpu > 0).sum()), int((tsdf == 255).sum())))
