"""Forward trunk and backward dgrad chain of the ResnetFC at the bench row count: fused.hip's 64-row ring kernels against wide.hip's
128-row kernels.  Kernel times come from the in-library HIP-event table (per launch, on the launch stream); results are compared.
usage: wide_probe.py [M] [reps]     env: PROBE_MASKS="1,1,1,3" (tile masks, repeated; default = the KITTI mix), PROBE_ZERO=1 (zero data)"""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
pat = [int(x) for x in os.environ.get("PROBE_MASKS", "1,1,1,3").split(",")]
dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16")
state = synth.mlp_state(1, 4)
params = [torch.as_tensor(state[n]).to(dev) for n in MLP_PARAM_NAMES]
ZERO = os.environ.get("PROBE_ZERO", "0") == "1"     # all-zero operands: the same instruction stream at the lowest switching power
if ZERO:
    params = [torch.zeros_like(t) for t in params]
pk = PackedMLP(params, 4, rcfg)
gen = torch.Generator().manual_seed(1)
ntile = (M + 127) // 128
masks = torch.tensor(pat, dtype=torch.uint8)[torch.arange(ntile) % len(pat)]
Z = (torch.randn(ntile * 128, 2480, generator=gen) * 0.5).to(torch.bfloat16).to(dev)
seg = [0]
for c, _, _ in rcfg.map_shapes():
    seg.append(seg[-1] + c)
for s_ in range(5):   # the gather writes exact zeros into the dense first 256 columns of tiles that miss a scale, nothing elsewhere
    act = ((masks.long() >> s_) & 1).bool().repeat_interleave(128).to(dev)
    Z[:, seg[s_]:seg[s_ + 1]] = torch.where(act[:, None], Z[:, seg[s_]:seg[s_ + 1]], torch.zeros((), dtype=torch.bfloat16, device=dev))
X = torch.randn(M, 48, generator=gen).clamp(-1, 1).to(dev)
X[:, 42:] = 0
dl = torch.randn(M, 4, generator=gen).to(dev)
if ZERO:
    Z.zero_(); X.zero_(); dl.zero_()
tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=dev)
tw = torch.zeros((M, 5, 4), device=dev)
st = torch.cuda.current_stream().cuda_stream
nzc = {m: sum(c for i, (c, _, _) in enumerate(rcfg.map_shapes()) if (m >> i) & 1) for m in set(pat)}
fl_f = sum(2.0 * min(128, M - t * 128) * 512 * (144 + 6 * 512 + 3 * nzc[int(masks[t])]) for t in range(ntile))
fl_b = 2.0 * M * 512 * 6 * 512


def timed(rows, name):
    r = rows.get(name)
    return r["total_ms"] * 1e3 / r["launches"] if r else float("nan")


res = {}
for name in ("ring", "wide_staged", "wide"):
    cfg = dataclasses.replace(rcfg, fused_min_rows=1, fwd_kernel=name.split("_")[0], bwd_kernel=name, wide_any_m=True, wgrad_overlap=True)
    cc = cfg.to_c()
    run = _MlpRun(M, 4, 1, dev)
    run.Z.copy_(Z); run.xenc.copy_(X); run.tile_mask[:ntile] = masks.to(dev)
    fwd = lambda: _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                                          C.byref(run.c), st), "fwd")
    gs = pk.grad_sink()
    dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=dev)
    dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
    bwd = lambda: _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                                           tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH.data_ptr(), dN.data_ptr(), None, st), "bwd")
    for _ in range(2):
        fwd(); bwd()
    torch.cuda.synchronize()
    lib.scenerf_hip_profile_enable(1)
    for _ in range(reps):
        fwd(); bwd()
    torch.cuda.synchronize()
    rows = {r["name"]: r for r in _capi.profile_collect()}
    tf, tb = timed(rows, "mlp_fwd_fused"), timed(rows, "mlp_bwd_fused")
    lib.scenerf_hip_profile_enable(0)
    print("%-5s M=%d masks=%s: forward %.1f us = %.0f TFLOP/s issued (%.1f %% of 2.5 PF) | dgrad chain %.1f us = %.0f TFLOP/s (%.1f %%)" % (
        name, M, pat, tf, fl_f / tf / 1e6, fl_f / tf / 1e6 / 25, tb, fl_b / tb / 1e6, fl_b / tb / 1e6 / 25), flush=True)
    print("      batched weight gradients %.1f us; linout_bwd %.1f us, linout_wgrad %.1f us" % (
        timed(rows, "gemm_wgrad_fc"), timed(rows, "linout_bwd"), timed(rows, "linout_wgrad")), flush=True)
    if name == "wide" and hasattr(lib, "scenerf_hip_test_wgrad_cyc"):
        buf = (C.c_ulonglong * (64 * 16))()
        lib.scenerf_hip_test_wgrad_cyc(buf, 64 * 16)
        import numpy as np
        a_ = np.array(buf[:], dtype=np.float64).reshape(64, 2, 8)
        for w_ in (0, 1):
            m_ = a_[:, w_, :].mean(0)
            print("      wgrad wave %d: per step: barrier %.0f  read0 %.0f  mma0 %.0f  read1 %.0f  mma1 %.0f  (steps %.0f) cycles" % (
                4 * w_, m_[0] / m_[5], m_[1] / m_[5], m_[2] / m_[5], m_[3] / m_[5], m_[4] / m_[5], m_[5]))
    res[name] = (run, dH, dN)
(a, dHa, dNa), (b, dHb, dNb) = res["ring"], res["wide"]
msg = ["logits max diff %.2e (scale %.2f)" % ((a.logits - b.logits).abs().max().item(), a.logits.abs().max().item())]
for i in range(4):
    x, y = a.H[i].float(), b.H[i].float()
    msg.append("H%d %.5f equal, max|diff|/max|x| %.1e" % (i, (x == y).float().mean().item(), ((x - y).abs().max() / x.abs().max()).item()))
for i in range(3):
    x, y = a.Nn[i].float(), b.Nn[i].float()
    msg.append("N%d %.5f equal" % (i, (x == y).float().mean().item()))
msg.append("sign bits equal %.6f" % (a.sign_bits[:6, :M] == b.sign_bits[:6, :M]).float().mean().item())
print("forward  wide vs ring: " + "; ".join(msg))
# the chains ran on different forward states (last-ulp differences): rerun the wide chain on the ring's state for a bit-exact check
run = a
cc = dataclasses.replace(rcfg, fused_min_rows=1, bwd_kernel="wide_staged", wide_any_m=True).to_c()
gs = pk.grad_sink()
dH2 = torch.zeros_like(dHa); dN2 = torch.zeros_like(dNa)
_capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                         tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH2.data_ptr(), dN2.data_ptr(), None, st), "bwd")
torch.cuda.synchronize()
print("backward wide vs ring on the same forward state: dH equal %s, dN equal %s" % (torch.equal(dHa, dH2), torch.equal(dNa, dN2)))
# ... and the default chain (lin_out's input gradient made in the prologue) against it
cc = dataclasses.replace(rcfg, fused_min_rows=1, bwd_kernel="wide", wide_any_m=True).to_c()
dH3 = torch.zeros_like(dHa); dN3 = torch.zeros_like(dNa)
_capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                         tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH3.data_ptr(), dN3.data_ptr(), None, st), "bwd")
torch.cuda.synchronize()
x, y = dHa[:, 1536:].float(), dH3[:, 1536:].float()
print("prologue dH3 vs linout_bwd dH3: %.3e of the elements differ, max |diff| / max |x| %.2e; dH0 rel L2 %.2e, equal %.5f" % (
    (x != y).float().mean().item(), ((x - y).abs().max() / x.abs().max()).item(),
    ((dHa[:, :512].float() - dH3[:, :512].float()).norm() / dHa[:, :512].float().norm()).item(), (dHa[:, :512] == dH3[:, :512]).float().mean().item()))
