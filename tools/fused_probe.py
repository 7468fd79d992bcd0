"""Time the ResnetFC forward (per-layer path and the fused kernels: ring / wide) on random inputs at the bench row count, and
compare the fused variants' outputs.   usage: fused_probe.py [M] [reps] [kernels: layers,ring,wide]
env: PROBE_MASK (scale mask of every tile, default 1 = KITTI's common case), PROBE_LEAN=1 (inference buffers), PROBE_COLD=1 (also time
calls behind a cache-flushing fill)"""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kernels = (sys.argv[3] if len(sys.argv) > 3 else "ring,wide").split(",")
maskv = int(os.environ.get("PROBE_MASK", "1"))
lean = bool(os.environ.get("PROBE_LEAN"))
dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16")
state = synth.mlp_state(1, 4)
params = [torch.as_tensor(state[n]).to(dev) for n in MLP_PARAM_NAMES]
pk = PackedMLP(params, 4, rcfg)
gen = torch.Generator().manual_seed(1)
Z = (torch.randn((M + 127) // 128 * 128, 2480, generator=gen) * 0.5).to(torch.bfloat16).to(dev)
X = torch.randn(M, 48, generator=gen).clamp(-1, 1).to(dev)
X[:, 42:] = 0
st = torch.cuda.current_stream().cuda_stream
res = {}
for name in kernels:
    cfg = dataclasses.replace(rcfg, fused_min_rows=-1) if name == "layers" else dataclasses.replace(rcfg, fused_min_rows=1, fwd_kernel=name, wide_any_m=True)
    cc = cfg.to_c()
    run = _MlpRun(M, 4, 1, dev, lean=lean and name != "layers")
    run.Z.copy_(Z); run.xenc.copy_(X); run.tile_mask.fill_(maskv)
    call = lambda: _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                                           C.byref(run.c), st), "fwd")
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if os.environ.get("PROBE_WARM_SWEEP"):   # start stagger of the first dispatch round (64-cycle units), alternated in one process
        import statistics
        vals = [int(x) for x in os.environ["PROBE_WARM_SWEEP"].split(",")]
        acc = {v: [] for v in vals}
        for rnd in range(5):
            for v in vals:
                lib.scenerf_hip_test_set_tuning(v, -1)
                call(); torch.cuda.synchronize()
                e0.record()
                for _ in range(reps):
                    call()
                e1.record()
                torch.cuda.synchronize()
                acc[v].append(e0.elapsed_time(e1) * 1e3 / reps)
        for v in vals:
            print("%-7s M=%d warm-up %d: median %.1f us (min %.1f)" % (name, M, v, statistics.median(acc[v]), min(acc[v])), flush=True)
        lib.scenerf_hip_test_set_tuning(1, -1)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    if os.environ.get("PROBE_COLD"):   # every call behind a 512-MB fill: weights and inputs come from HBM, as after a re-pack by another kernel
        flush = torch.empty(128 << 20, dtype=torch.float32, device=dev)
        tot = 0.0
        for _ in range(reps):
            flush.zero_()
            e0.record(); call(); e1.record()
            torch.cuda.synchronize()
            tot += e0.elapsed_time(e1)
        print("%-7s cold: %.3f ms per forward" % (name, tot / reps), flush=True)
    nz = sum(c for i, (c, _, _) in enumerate(rcfg.map_shapes()) if (maskv >> i) & 1)
    fl = 2.0 * M * 512 * (144 + 6 * 512 + 3 * nz)
    print("%-7s M=%d mask=%d%s: %.3f ms per forward  %.0f TFLOP/s issued = %.1f %% of 2.5 PF" % (
        name, M, maskv, " lean" if lean else "", ms, fl / ms / 1e9, fl / ms / 1e9 / 25), flush=True)
    res[name] = run
ref = res.get("ring")
for name, run in res.items():
    if run is ref or ref is None or name == "layers":
        continue
    lg = (run.logits - ref.logits).abs().max().item(), ref.logits.abs().max().item()
    msg = ["logits max diff %.2e (scale %.2f)" % lg]
    if not lean:
        for i in range(4):
            a, b = ref.H[i].float(), run.H[i].float()
            msg.append("H%d: %.5f equal, max |diff| / max|x| %.1e" % (i, (a == b).float().mean().item(), ((a - b).abs().max() / a.abs().max()).item()))
        for i in range(3):
            a, b = ref.Nn[i].float(), run.Nn[i].float()
            msg.append("N%d: %.5f equal" % (i, (a == b).float().mean().item()))
        msg.append("sign bits equal %.6f" % (ref.sign_bits[:6, :M] == run.sign_bits[:6, :M]).float().mean().item())
    print("%s vs %s: %s" % (name, "ring", "; ".join(msg)))
