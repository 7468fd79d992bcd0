"""HBM roofline of the compositing kernels at inference-sized chunks (through the C ABI, HIP-event timed)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenerf_amd import _capi
lib = _capi.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
for (R, N) in [(1200, 128), (8192, 64), (65536, 64), (65536, 128), (262144, 128), (65536, 512)]:
    logits = torch.randn(R * N, 4, device=dev); logits[:, 3] -= 2
    dist = torch.sort(torch.rand(R, N, device=dev) * 100 + 0.1, dim=1).values
    z = dist * 0.97
    f = lambda *s: torch.empty(s, device=dev)
    dens, al, w, dep, col, clo, wat = f(R, N), f(R, N), f(R, N), f(R), f(R, 3), f(R), f(R)
    ci = torch.empty(R, dtype=torch.int32, device=dev)
    gd, gc = torch.randn(R, device=dev), torch.randn(R, 3, device=dev)
    dl, dd, dz = f(R * N, 4), f(R, N), f(R, N)
    def fwd():
        lib.scenerf_hip_composite_forward(logits.data_ptr(), dist.data_ptr(), z.data_ptr(), R, N, dens.data_ptr(), al.data_ptr(), w.data_ptr(),
                                          dep.data_ptr(), col.data_ptr(), clo.data_ptr(), wat.data_ptr(), ci.data_ptr(), st)
    def bwd():
        lib.scenerf_hip_composite_backward(logits.data_ptr(), dist.data_ptr(), z.data_ptr(), R, N, gd.data_ptr(), gc.data_ptr(), None, None, None,
                                           None, dl.data_ptr(), dd.data_ptr(), dz.data_ptr(), st)
    res = []
    for fn, bytes_per_ray in ((fwd, 32 * N + 24), (bwd, 48 * N + 40)):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        res.append((us, R * bytes_per_ray / us / 1e3))
    print("R=%7d N=%3d  fwd %8.1f us %7.1f GB/s (%.1f%% of 8 TB/s)   bwd %8.1f us %7.1f GB/s (%.1f%%)" % (
        R, N, res[0][0], res[0][1], res[0][1] / 80, res[1][0], res[1][1], res[1][1] / 80))
