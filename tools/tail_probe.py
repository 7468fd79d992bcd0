"""The per-ray tail the PRODUCT launches (scenerf_hip_ray_tail_forward / _backward: compositing + RaySOM, their autograd + the sampler's)
at inference-sized chunks, through the C ABI, HIP-event timed, against the compositing pass's algorithmic bytes: with the RaySOM
half (training), without it (loss_kl = NULL: what a depth / colour render under no_grad launches -- all (R, N) outputs, and depth + colour
only), and the backward.  Run under rocprofv3 --pmc by tools/profile_tail.sh for the counter bytes of the same launches.
usage: tail_probe.py [R N]..."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from scenerf_amd import _capi
from scenerf_amd.config import RenderConfig

lib = _capi.load()
dev = "cuda"
st = torch.cuda.current_stream().cuda_stream
args = [int(a) for a in sys.argv[1:]]
shapes = list(zip(args[0::2], args[1::2])) or [(65536, 128), (4096, 512)]
out = []
for (R, N) in shapes:
    P = N // 8
    U = N - 4 * P
    cc = RenderConfig.kitti(n_pts_uni=U, n_pts_per_gaussian=P).to_c()
    assert cc.n_samples == N
    G = 4
    logits = torch.randn(R * N, 4, device=dev); logits[:, 3] -= 2
    dist = torch.sort(torch.rand(R, N, device=dev) * 100 + 0.1, dim=1).values
    z = dist * 0.97
    gm = torch.sort(torch.rand(R, G, device=dev) * 80 + 2, dim=1).values
    gs = torch.rand(R, G, device=dev) * 4 + 1.5
    perm = torch.argsort(torch.rand(R, N, device=dev), dim=1).to(torch.int32)
    offs, anchors = torch.randn(R, G, 2, device=dev), torch.linspace(12.5, 87.5, G, device=dev)
    noise, unit = torch.randn(R, G * P, device=dev), torch.nn.functional.normalize(torch.randn(R, 3, device=dev), dim=1)
    f = lambda *s: torch.empty(s, device=dev)
    dens, al, w, dep, col, clo, wat = f(R, N), f(R, N), f(R, N), f(R), f(R, 3), f(R), f(R)
    ci = torch.empty(R, dtype=torch.int32, device=dev)
    lk, sm, sv, ks = f(R), f(R, G), f(R, G), f(R, G, 3)
    gd, gc = torch.randn(R, device=dev), torch.randn(R, 3, device=dev)
    dl, do = f(R * N, 4), f(R, G, 2)
    p = lambda t: t.data_ptr()
    som = lambda: lib.scenerf_hip_ray_tail_forward(C.byref(cc), p(logits), p(dist), p(z), p(gm), p(gs), R, p(dens), p(al), p(w), p(dep), p(col),
                                                   p(clo), p(wat), p(ci), p(lk), p(sm), p(sv), p(ks), None, st)
    nosom_all = lambda: lib.scenerf_hip_ray_tail_forward(C.byref(cc), p(logits), p(dist), p(z), None, None, R, p(dens), p(al), p(w), p(dep), p(col),
                                                         p(clo), p(wat), p(ci), None, None, None, None, None, st)
    nosom_dc = lambda: lib.scenerf_hip_ray_tail_forward(C.byref(cc), p(logits), p(dist), p(z), None, None, R, None, None, None, p(dep), p(col),
                                                        p(clo), p(wat), p(ci), None, None, None, None, None, st)
    bwd = lambda: lib.scenerf_hip_ray_tail_backward(C.byref(cc), p(logits), p(dist), p(z), R, p(gd), p(gc), None, None, None, None, p(offs), p(anchors),
                                                    p(noise), p(unit), p(gm), p(gs), p(perm), p(ks), None, None, None, p(dl), p(do), None, None, None,
                                                    None, None, st)
    for name, fn, bpr in (("tail_fwd (compositing + RaySOM, all outputs)", som, 32 * N + 24),
                          ("tail_fwd_nosom (compositing, all outputs)", nosom_all, 32 * N + 24),
                          ("tail_fwd_nosom (compositing, depth + colour only)", nosom_dc, 20 * N + 24),
                          ("tail_bwd (compositing + sampler / KL backward)", bwd, 44 * N + 40)):
        for _ in range(3):
            assert fn() == 0, lib.scenerf_hip_last_error()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        gbs = R * bpr / us / 1e3
        out.append(dict(rays=R, samples=N, kernel=name, algorithmic_bytes_per_launch=R * bpr, avg_launch_us=round(us, 1), GBps=round(gbs, 1),
                        frac_of_8TBps=round(gbs / 8000, 4)))
        print("R=%6d N=%3d  %-52s %8.1f us  %7.1f GB/s  %.3f of 8 TB/s   (%.1f MB algorithmic)" % (R, N, name, us, gbs, gbs / 8000, R * bpr / 1e6))
os.makedirs("gpurun_out", exist_ok=True) if os.path.isdir(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")) else None
if os.environ.get("TAIL_PROBE_JSON"):
    json.dump(out, open(os.environ["TAIL_PROBE_JSON"], "w"), indent=1)
