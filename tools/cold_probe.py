"""ray_tail_fwd at R = 1200 warm (back to back) and cold (256 MB of other traffic between launches: L2 / instruction cache evicted)."""
import ctypes as C, sys, torch
sys.path.insert(0, '.')
import bench
from scenerf_amd import _capi
from scenerf_amd.config import RenderConfig
lib = _capi.load(); dev = "cuda"
R, N, G = 1200, 128, 4
cc = RenderConfig.kitti(n_pts_uni=64, n_pts_per_gaussian=16).to_c()
st = torch.cuda.current_stream().cuda_stream
logits = torch.randn(R * N, 4, device=dev); logits[:, 3] -= 2
dist = torch.sort(torch.rand(R, N, device=dev) * 100 + 0.1, dim=1).values
z = dist * 0.97
gm = torch.sort(torch.rand(R, G, device=dev) * 80 + 2, dim=1).values
gs = torch.rand(R, G, device=dev) * 4 + 1.5
f = lambda *s: torch.empty(s, device=dev)
dens, al, w, dep, col, clo, wat = f(R, N), f(R, N), f(R, N), f(R), f(R, 3), f(R), f(R)
ci = torch.empty(R, dtype=torch.int32, device=dev)
lk, sm, sv, ks = f(R), f(R, G), f(R, G), f(R, G, 3)
fwd = lambda: lib.scenerf_hip_ray_tail_forward(C.byref(cc), logits.data_ptr(), dist.data_ptr(), z.data_ptr(), gm.data_ptr(), gs.data_ptr(), R,
                                               dens.data_ptr(), al.data_ptr(), w.data_ptr(), dep.data_ptr(), col.data_ptr(), clo.data_ptr(),
                                               wat.data_ptr(), ci.data_ptr(), lk.data_ptr(), sm.data_ptr(), sv.data_ptr(), ks.data_ptr(), None, st)
cfwd = lambda: lib.scenerf_hip_composite_forward(logits.data_ptr(), dist.data_ptr(), z.data_ptr(), R, N, dens.data_ptr(), al.data_ptr(),
                                                 w.data_ptr(), dep.data_ptr(), col.data_ptr(), clo.data_ptr(), wat.data_ptr(), ci.data_ptr(), st)
big = torch.empty(512 * 1024 * 1024 // 4, device=dev)
for name, fn in (("ray_tail_fwd", fwd), ("composite_fwd", cfwd)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    for mode in ("warm", "cold"):
        ts = []
        for _ in range(8):
            if mode == "cold":
                big.add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        print("%-14s %s: %s us" % (name, mode, " ".join("%.1f" % t for t in ts)))
