"""Where a wide.hip block's cycles go: per-block s_memtime stamps of wave 0 (library built with SRF_LIB_TAG=cyc SRF_EXTRA_FLAGS=-DH_CYC).
usage: SRF_LIB_TAG=cyc python tools/wide_cycles.py [M]      env PROBE_MASKS, PROBE_ZERO as in wide_probe.py"""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

M = int(sys.argv[1]) if len(sys.argv) > 1 else 153600
pat = [int(x) for x in os.environ.get("PROBE_MASKS", "1,1,1,3").split(",")]
dev = torch.device("cuda:0")
lib = _capi.load()
lib.scenerf_hip_test_wide_cyc.argtypes = [C.c_void_p]
rcfg = RenderConfig.kitti(precision="bf16")
state = synth.mlp_state(1, 4)
ZERO = os.environ.get("PROBE_ZERO", "0") == "1"    # all-zero weights and inputs: the same instruction stream at the lowest switching power
pk = PackedMLP([torch.zeros_like(torch.as_tensor(state[n])).to(dev) if ZERO else torch.as_tensor(state[n]).to(dev) for n in MLP_PARAM_NAMES], 4, rcfg)
gen = torch.Generator().manual_seed(1)
ntile = (M + 127) // 128
masks = torch.tensor(pat, dtype=torch.uint8)[torch.arange(ntile) % len(pat)]
cfg = dataclasses.replace(rcfg, fused_min_rows=1, fwd_kernel="wide", bwd_kernel="wide", wide_any_m=True)
cc = cfg.to_c()
run = _MlpRun(M, 4, 1, dev)
run.Z.copy_((torch.randn(ntile * 128, 2480, generator=gen) * 0.5).to(torch.bfloat16).to(dev))
X = torch.randn(M, 48, generator=gen).clamp(-1, 1); X[:, 42:] = 0
run.xenc.copy_(X.to(dev)); run.tile_mask[:ntile] = masks.to(dev)
dl = torch.randn(M, 4, generator=gen).to(dev)
if ZERO:
    run.Z.zero_(); X.zero_(); dl.zero_()
tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=dev); tw = torch.zeros((M, 5, 4), device=dev)
st = torch.cuda.current_stream().cuda_stream
gs = pk.grad_sink()
dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=dev); dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
fwd = lambda: _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M, C.byref(run.c), st), "fwd")
bwd = lambda: _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                                       tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH.data_ptr(), dN.data_ptr(), None, st), "bwd")
fwd(); bwd(); torch.cuda.synchronize()
nb = ntile
for name, call, labels in (("forward", fwd, ["setup", "dma0"] + sum([["K%d" % l, "wait%d" % l, "epi%d" % l] for l in range(7)], []) + ["stream-out", "lin_out"]),
                           ("backward", None, ["setup", "stage dH3"] + sum([["K%d" % l, "wait%d" % l, "epi%d" % l] for l in range(6)], []) + ["stream-out", "end"])):
    buf = torch.zeros((nb, 64), dtype=torch.int64, device=dev)
    assert lib.scenerf_hip_test_wide_cyc(buf.data_ptr()) == 0
    if call is None:
        lib.scenerf_hip_test_wide_cyc(None)      # the backward entry runs other kernels first; only its chain kernel must stamp
        fwd(); torch.cuda.synchronize()
        lib.scenerf_hip_test_wide_cyc(buf.data_ptr())
        bwd()
    else:
        call()
    torch.cuda.synchronize()
    lib.scenerf_hip_test_wide_cyc(None)
    t = buf.cpu().double()
    n = int((t[0] != 0).sum())
    d = t[:, 1:n] - t[:, :n - 1]
    tot = t[:, n - 1] - t[:, 0]
    span = (t[:, n - 1].max() - t[:, 0].min()).item()
    print("%s: %d stamps, block total mean %.0f cycles (min %.0f max %.0f); first start -> last end %.0f cycles (= %.2f blocks deep)" % (
        name, n, tot.mean().item(), tot.min().item(), tot.max().item(), span, span / tot.mean().item()))
    # per dispatch round (blocks are handed out in id order, 256 at a time): start offset from the kernel's first stamp, mean length --
    # do the first round's blocks (every CU in the same phase at the same time) run slower than the later, desynchronised ones?
    # (s_memtime is per XCD: block b runs on XCD b % 8; offsets are taken against the first stamp of the same XCD)
    xcd = torch.arange(nb) % 8
    t0 = torch.stack([t[xcd == x, 0].min() for x in range(8)])[xcd]
    st_, en_ = t[:, 0] - t0, t[:, n - 1] - t0
    print("  kernel span per XCD (first start -> last end): " + " ".join("%.0f" % en_[xcd == x].max().item() for x in range(8)))
    for r_ in range((nb + 255) // 256):
        sl = slice(256 * r_, min(nb, 256 * r_ + 256))
        print("  round %d (%d blocks): start %.0f..%.0f, length mean %.0f (min %.0f max %.0f), end mean %.0f max %.0f" % (
            r_, sl.stop - sl.start, st_[sl].min().item(), st_[sl].max().item(), tot[sl].mean().item(), tot[sl].min().item(),
            tot[sl].max().item(), en_[sl].mean().item(), en_[sl].max().item()))
    for m_ in sorted(set(pat)):
        sel = (masks == m_)
        print("  mask %d (%d blocks): " % (m_, int(sel.sum())) + "  ".join("%s %.0f" % (labels[i] if i < len(labels) else "?", d[sel, i].mean().item()) for i in range(n - 1)))
