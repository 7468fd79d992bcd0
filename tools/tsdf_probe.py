"""TSDF fusion throughput: frames/s and the HBM rate of scenerf_hip_tsdf_integrate on a scene-completion sized volume,
next to the numpy oracle (the reference's CPU path restated) on the host.  usage: tsdf_probe.py [voxel_size]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from scenerf_amd.fusion import TSDFVolume
import tsdf_scene, tsdf_oracle as orc

vs = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
sc = tsdf_scene.make(seed=7, im_h=480, im_w=640, n_frames=3)
for sem in ("gpu", "cpu"):
    vol = TSDFVolume(sc["vol_bnds"].copy(), voxel_size=vs, trunc_margin=sc["trunc_margin"], semantics=sem)
    n = int(np.prod(vol._vol_dim))
    fr = sc["frames"]
    vol.integrate(fr[0]["color"], fr[0]["depth"], sc["cam_intr"], fr[0]["pose"])
    torch.cuda.synchronize()
    reps = 10
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for i in range(reps):
        f = fr[i % len(fr)]
        vol.integrate(f["color"], f["depth"], sc["cam_intr"], f["pose"])
    e1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    upd = int((vol._weight > 0).sum())
    ms = e0.elapsed_time(e1) / reps
    # the kernel alone (in-library HIP events); per-frame voxels in the frustum for the traffic figure
    from scenerf_amd import _capi
    lib = _capi.load()
    lib.scenerf_hip_profile_enable(1)
    before = vol._weight.clone()
    vol.integrate(fr[0]["color"], fr[0]["depth"], sc["cam_intr"], fr[0]["pose"])
    torch.cuda.synchronize()
    k = [r for r in _capi.profile_collect() if r["name"] == "tsdf_integrate"][0]
    lib.scenerf_hip_profile_enable(0)
    touched = int((vol._weight != before).sum())
    kus = k["total_ms"] * 1e3 / k["launches"]
    print("   kernel alone: %.1f us for %.1f M voxels (%.1f M updated) -> %.0f G voxels/s ; algorithmic volume traffic %.0f MB (24 B per "
          "updated voxel: three fp32 volumes read + written) = %.2f TB/s" % (kus, n / 1e6, touched / 1e6, n / kus / 1e3, touched * 24 / 1e6,
                                                                            touched * 24 / kus / 1e6))
# host: the oracle (vectorised numpy, what the reference does without pycuda) on a 64x smaller volume
sc2 = tsdf_scene.make(seed=7, im_h=480, im_w=640, n_frames=1)
dim, origin, t, w, c = orc.new_volume(sc2["vol_bnds"], vs * 4)
f = sc2["frames"][0]
t0 = time.perf_counter()
orc.integrate_cpu_semantics(t, w, c, origin, vs * 4, orc.fold_color(f["color"]), f["depth"], sc2["cam_intr"], f["pose"], sc2["trunc_margin"])
dt = time.perf_counter() - t0
print("host numpy oracle (cpu semantics): %.2f M voxels in %.2f s -> %.4f G voxels/s" % (t.size / 1e6, dt, t.size / dt / 1e9))
