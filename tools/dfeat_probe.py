"""Feature-gradient kernel (dfeat.hip) against the GEMM family's scatter epilogue at the bench shape: time and result.
usage: dfeat_probe.py [M]   env PROBE_MASKS as in wide_probe.py"""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP

M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
pat = [int(x) for x in os.environ.get("PROBE_MASKS", "1,1,1,3").split(",")]
dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16")
state = synth.mlp_state(1, 4)
pk = PackedMLP([torch.as_tensor(state[n]).to(dev) for n in MLP_PARAM_NAMES], 4, rcfg)
gen = torch.Generator().manual_seed(3)
ntile = (M + 127) // 128
masks = torch.tensor(pat, dtype=torch.uint8)[torch.arange(ntile) % len(pat)].to(dev)
dH = (torch.randn(M, 2048, generator=gen) * 0.1).to(torch.bfloat16).to(dev)
shapes = rcfg.map_shapes()
# taps: per ray (128 rows) a slowly advancing texel walk, 2x2 neighbours, like an epipolar curve
tex = torch.full((M, 5, 4), -1, dtype=torch.int32)
tw = torch.zeros((M, 5, 4))
r = torch.arange(M)
for s_, (c, h, w) in enumerate(shapes[:2]):
    x0 = (torch.randint(0, w - 40, (ntile,), generator=gen).repeat_interleave(128)[:M] + (r % 128) * 30 // 128)
    y0 = torch.randint(0, h - 2, (ntile,), generator=gen).repeat_interleave(128)[:M]
    act = ((masks.cpu().long() >> s_) & 1).bool().repeat_interleave(128)[:M]
    for k, (dx, dy) in enumerate(((0, 0), (1, 0), (0, 1), (1, 1))):
        tex[:, s_, k] = torch.where(act, ((y0 + dy) * w + x0 + dx).int(), torch.full((M,), -1, dtype=torch.int32))
        tw[:, s_, k] = 0.25
tex, tw = tex.to(dev), tw.to(dev)
st = torch.cuda.current_stream().cuda_stream
res = {}
for name in ("gemm", "dfeat"):
    cc = dataclasses.replace(rcfg, dfeat_gemm=(name == "gemm")).to_c()
    gm = [torch.zeros((h, w, c), device=dev) for (c, h, w) in shapes]
    arr = (C.c_void_p * 5)(*[g.data_ptr() if i < 2 else None for i, g in enumerate(gm)])
    call = lambda: _capi.check(lib.scenerf_hip_mlp_feature_grads(C.byref(cc), C.byref(pk.c), masks.data_ptr(), tex.data_ptr(), tw.data_ptr(), M,
                                                                 dH.data_ptr(), arr, st), "feature_grads")
    call(); torch.cuda.synchronize()
    for g in gm:
        g.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    call(); torch.cuda.synchronize()
    res[name] = [g.clone() for g in gm[:2]]
    e0.record()
    for _ in range(10):
        call()
    e1.record(); torch.cuda.synchronize()
    print("%-6s M=%d masks=%s: %.1f us per launch" % (name, M, pat, e0.elapsed_time(e1) * 100), flush=True)
for i in range(2):
    a, b = res["gemm"][i], res["dfeat"][i]
    print("level %d: max |diff| %.3e  (scale %.3e)%s" % (i, (a - b).abs().max().item(), a.abs().max().item(), "" if bool(torch.isfinite(b).all()) else "  NON-FINITE"))

if os.environ.get("SRF_LIB_TAG") and hasattr(lib, "scenerf_hip_test_dfeat_cyc"):   # a -DH_CYC build: where a workgroup's cycles go (3-tile kernel: start, K loop, table, rounds)
    lib.scenerf_hip_test_dfeat_cyc.argtypes = [C.c_void_p]
    buf = torch.zeros((ntile, 16), dtype=torch.int64, device=dev)
    lib.scenerf_hip_test_dfeat_cyc(buf.data_ptr())
    cc = dataclasses.replace(rcfg, dfeat_gemm=False).to_c()
    call(); torch.cuda.synchronize()
    lib.scenerf_hip_test_dfeat_cyc(None)
    t = buf.cpu().double()
    for m_ in sorted(set(pat)):
        sel = (masks.cpu() == m_)
        tt = t[sel][:, :12]
        ph = t[sel][:, 12:].mean(0).tolist()     # K-loop phases of wave 0, summed over the steps (the launch that ran last on this block)
        print("mask %d: K loop of wave 0: waiting for the step's pieces %.0f, at the barrier %.0f, issuing the next pieces %.0f, reads + MFMAs %.0f cycles" % (m_, *ph))
        n = int((tt[0] != 0).sum())
        d = tt[:, 1:n] - tt[:, :n - 1]
        print("mask %d: %d stamps; mean cycles between stamps: %s ; total %.0f" % (m_, n, " ".join("%.0f" % x for x in d.mean(0).tolist()), (tt[:, n - 1] - tt[:, 0]).mean().item()))
