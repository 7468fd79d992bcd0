"""Layout-change kernels at the KITTI map shapes: (C,H,W) fp32 -> (H,W,C) bf16 (forward) and (H,W,C) fp32 -> (C,H,W) fp32 (map gradients).
Kernel times from the in-library HIP-event table."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi
from scenerf_amd.config import RenderConfig
lib = _capi.load()
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
rcfg = RenderConfig.kitti(precision="bf16")
shapes = [s for i, s in enumerate(rcfg.map_shapes()) if i not in rcfg.direct_scales]
tot_f = tot_b = 0.0
for (c, h, w) in shapes:
    x = torch.randn(c, h, w, device=dev)
    y = torch.empty(h, w, c, dtype=torch.bfloat16, device=dev)
    g = torch.randn(h, w, c, device=dev)
    gc = torch.empty(c, h, w, device=dev)
    for _ in range(2):
        lib.scenerf_hip_maps_chw_to_hwc(x.data_ptr(), y.data_ptr(), c, h, w, 1, st)
        lib.scenerf_hip_grads_hwc_to_chw(g.data_ptr(), gc.data_ptr(), c, h, w, st)
    torch.cuda.synchronize()
    assert torch.equal(y.float(), x.permute(1, 2, 0).to(torch.bfloat16).float()) and torch.equal(gc, g.permute(2, 0, 1))
    lib.scenerf_hip_profile_enable(1)
    for _ in range(10):
        lib.scenerf_hip_maps_chw_to_hwc(x.data_ptr(), y.data_ptr(), c, h, w, 1, st)
        lib.scenerf_hip_grads_hwc_to_chw(g.data_ptr(), gc.data_ptr(), c, h, w, st)
    torch.cuda.synchronize()
    rows = {r["name"]: r for r in _capi.profile_collect()}
    lib.scenerf_hip_profile_enable(0)
    f = rows["maps_chw_to_hwc"]["total_ms"] * 100
    b = rows["grads_hwc_to_chw"]["total_ms"] * 100
    n = c * h * w
    print("(%d,%d,%d): CHW fp32 -> HWC bf16 %.1f us (%.2f TB/s) | HWC fp32 -> CHW fp32 %.1f us (%.2f TB/s)" % (c, h, w, f, n * 6 / f / 1e6, b, n * 8 / b / 1e6))
    tot_f += f; tot_b += b
print("total forward %.1f us, backward %.1f us" % (tot_f, tot_b))
