"""Time the ResnetFC forward (layer path vs fused kernel) on random inputs at the bench row count.
usage: fused_probe.py [M] [reps] [mode: both|fused|layers]"""
import ctypes as C, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "both"
maskv = int(os.environ.get("PROBE_MASK", "7"))
dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16")
cc = rcfg.to_c()
state = synth.mlp_state(1, 4)
params = [torch.as_tensor(state[n]).to(dev) for n in MLP_PARAM_NAMES]
pk = PackedMLP(params, 4, rcfg)
run = _MlpRun(M, 4, 1, dev, keep_acts=not os.environ.get("PROBE_LEAN"))   # PROBE_LEAN=1: inference buffers (no activation / sign-bit writes)
run.Z.copy_(torch.randn(run.Z.shape, device=dev) * 0.5)
run.xenc.copy_(torch.randn(run.xenc.shape, device=dev).clamp(-1, 1))
run.tile_mask.fill_(maskv)
st = torch.cuda.current_stream().cuda_stream
for name, env in (("layers", str(1 << 30)), ("fused", "0")):
    if mode not in ("both", name):
        continue
    os.environ["SRF_FUSED_MIN_M"] = env
    for _ in range(3):
        _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                                C.byref(run.c), st), "fwd")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M,
                                                C.byref(run.c), st), "fwd")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    nz = sum(c for i, (c, _, _) in enumerate(rcfg.map_shapes()) if (maskv >> i) & 1)
    fl = 2.0 * M * 512 * (144 + 6 * 512 + 3 * nz)
    print("%-7s M=%d mask=%d: %.3f ms per forward  (%.0f TFLOP/s useful)" % (name, M, maskv, ms, fl / ms / 1e9))
