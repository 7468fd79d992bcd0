"""The 128-row forward trunk and dgrad chain alone at the bench row count: in-library HIP events per launch (one line, for alternating
A/B runs of variant libraries: `SRF_LIB_TAG=<tag> python tools/wide_time.py [M] [reps]`).  env PROBE_MASKS as in wide_probe.py;
PROBE_CHECK=<file>: save (first run) / compare (later runs) the forward's logits + saved activations and the chain's dH / dN."""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
pat = [int(x) for x in os.environ.get("PROBE_MASKS", "1,1,1,3").split(",")]
dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16")
state = synth.mlp_state(1, 4)
pk = PackedMLP([torch.as_tensor(state[n]).to(dev) for n in MLP_PARAM_NAMES], 4, rcfg)
gen = torch.Generator().manual_seed(1)
ntile = (M + 127) // 128
masks = torch.tensor(pat, dtype=torch.uint8)[torch.arange(ntile) % len(pat)]
Z = (torch.randn(ntile * 128, 2480, generator=gen) * 0.5).to(torch.bfloat16).to(dev)
seg = [0]
for c, _, _ in rcfg.map_shapes():
    seg.append(seg[-1] + c)
for s_ in range(5):
    act = ((masks.long() >> s_) & 1).bool().repeat_interleave(128).to(dev)
    Z[:, seg[s_]:seg[s_ + 1]] = torch.where(act[:, None], Z[:, seg[s_]:seg[s_ + 1]], torch.zeros((), dtype=torch.bfloat16, device=dev))
X = torch.randn(M, 48, generator=gen).clamp(-1, 1).to(dev)
X[:, 42:] = 0
dl = torch.randn(M, 4, generator=gen).to(dev)
tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=dev)
tw = torch.zeros((M, 5, 4), device=dev)
st = torch.cuda.current_stream().cuda_stream
nzc = {m: sum(c for i, (c, _, _) in enumerate(rcfg.map_shapes()) if (m >> i) & 1) for m in set(pat)}
fl_f = sum(2.0 * min(128, M - t * 128) * 512 * (144 + 6 * 512 + 3 * nzc[int(masks[t])]) for t in range(ntile))
fl_b = 2.0 * M * 512 * (6 * 512 + 48)
cfg = dataclasses.replace(rcfg, fused_min_rows=1, fwd_kernel="wide", bwd_kernel="wide", wide_any_m=True, wgrad_overlap=True)
cc = cfg.to_c()
run = _MlpRun(M, 4, 1, dev)
run.Z.copy_(Z); run.xenc.copy_(X); run.tile_mask[:ntile] = masks.to(dev)
gs = pk.grad_sink()
dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=dev)
dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
fwd = lambda: _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                                      C.byref(run.c), st), "fwd")
bwd = lambda: _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                                       tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH.data_ptr(), dN.data_ptr(), None, st), "bwd")
for _ in range(3):
    fwd(); bwd()
torch.cuda.synchronize()
if os.environ.get("PROBE_WARM_SWEEP"):   # "0,4,8,12,...": the kernels' start stagger (64-cycle units per XCD position), alternated in ONE process
    import statistics
    vals = [int(x) for x in os.environ["PROBE_WARM_SWEEP"].split(",")]
    acc = {v: ([], []) for v in vals}
    for rnd in range(int(os.environ.get("PROBE_ROUNDS", "4"))):
        for v in vals:
            lib.scenerf_hip_test_set_tuning(v, -1)
            fwd(); bwd(); torch.cuda.synchronize()
            lib.scenerf_hip_profile_enable(1)
            for _ in range(reps // 2):
                fwd(); bwd()
            torch.cuda.synchronize()
            rows = {r["name"]: r for r in _capi.profile_collect()}
            lib.scenerf_hip_profile_enable(0)
            acc[v][0].append(rows["mlp_fwd_fused"]["total_ms"] * 1e3 / rows["mlp_fwd_fused"]["launches"])
            acc[v][1].append(rows["mlp_bwd_fused"]["total_ms"] * 1e3 / rows["mlp_bwd_fused"]["launches"])
    for v in vals:
        print("warm-up %d: forward median %.1f us (min %.1f) | dgrad chain median %.1f us (min %.1f)" % (
            v, statistics.median(acc[v][0]), min(acc[v][0]), statistics.median(acc[v][1]), min(acc[v][1])), flush=True)
    sys.exit(0)
if os.environ.get("PROBE_LOOP_S"):     # keep the GPU busy for a while (tools/clock_probe.sh reads clocks and power meanwhile)
    import time
    t_end = time.time() + float(os.environ["PROBE_LOOP_S"])
    which = os.environ.get("PROBE_LOOP_WHAT", "fwd")
    while time.time() < t_end:
        for _ in range(50):
            if which != "bwd":
                fwd()
            if which != "fwd":
                bwd()
        torch.cuda.synchronize()
lib.scenerf_hip_profile_enable(1)
for _ in range(reps):
    fwd(); bwd()
torch.cuda.synchronize()
rows = {r["name"]: r for r in _capi.profile_collect()}
lib.scenerf_hip_profile_enable(0)
us = lambda n: rows[n]["total_ms"] * 1e3 / rows[n]["launches"] if n in rows else float("nan")
tf, tb = us("mlp_fwd_fused"), us("mlp_bwd_fused")
msg = ""
chk = os.environ.get("PROBE_CHECK")
if chk:
    cur = {"logits": run.logits, "H3": run.H[3], "N1": run.Nn[1], "sign": run.sign_bits, "dH": dH, "dN": dN}
    if os.path.exists(chk):
        ref = torch.load(chk)
        msg = " | vs %s: " % os.path.basename(chk) + " ".join(
            "%s %s" % (k, "==" if torch.equal(ref[k], v.cpu()) else "max|d| %.2e" % (ref[k].float() - v.cpu().float()).abs().max().item()) for k, v in cur.items())
    else:
        torch.save({k: v.cpu() for k, v in cur.items()}, chk)
        msg = " | saved %s" % chk
print("lib %-8r M=%d: forward %.1f us (%.3f of 2.5 PF) | dgrad chain %.1f us (%.3f) | wgrad batch %.1f us | linout wgrad %.1f us%s" % (
    os.environ.get("SRF_LIB_TAG", ""), M, tf, fl_f / tf / 1e6 / 2500, tb, fl_b / tb / 1e6 / 2500, us("gemm_wgrad_fc"), us("linout_wgrad"), msg), flush=True)
