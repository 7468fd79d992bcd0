"""Run-to-run reproducibility of the parameter gradients of scenerf_hip_mlp_backward on identical inputs (fused chain and per-layer path):
anything beyond fp32 atomic-ordering noise (~1e-6) between two runs of the SAME path is a race."""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16", sphere_W=376, sphere_H=114)
state = synth.mlp_state(102, 4)
params = [state[n].to(dev) for n in MLP_PARAM_NAMES]
pk = PackedMLP(params, 4, rcfg)
st = torch.cuda.current_stream().cuda_stream
for M in (4133, 40000):
    gen = torch.Generator().manual_seed(M + 1)
    run = _MlpRun(M, 4, 1, dev)
    run.Z.copy_((torch.randn(run.Z.shape, generator=gen) * 0.5).to(torch.bfloat16).to(dev))
    xe = torch.zeros((M, 48)); xe[:, :42] = torch.randn(M, 42, generator=gen).clamp(-1, 1)
    run.xenc.copy_(xe.to(dev))
    run.tile_mask.fill_(31)
    cc = rcfg.to_c()
    _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M, C.byref(run.c), st), "fwd")
    dl = torch.randn(M, 4, generator=gen).to(dev)
    tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=dev); tw = torch.zeros((M, 5, 4), device=dev)
    for name in ("layers", "fused"):
        cc = dataclasses.replace(rcfg, fused_backward=(name == "fused")).to_c()
        outs = []
        for rep in range(6):
            gs = pk.grad_sink(); pk.gflat.zero_()
            dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=dev); dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
            _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                                     tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH.data_ptr(), dN.data_ptr(), None, st), "bwd")
            torch.cuda.synchronize()
            outs.append([g.clone() for g in pk.unpack_grads()])
        worst = {}
        for rep in range(1, 6):
            for n, a, b in zip(MLP_PARAM_NAMES, outs[0], outs[rep]):
                r = float((a - b).norm() / max(float(a.norm()), 1e-20))
                worst[n] = max(worst.get(n, 0.0), r)
        bad = {k: "%.1e" % v for k, v in worst.items() if v > 1e-5}
        print("M=%d %s: run-to-run rel L2 > 1e-5: %s   (max %.1e)" % (M, name, bad, max(worst.values())))

# where does lin_in.bias deviate?  reference = column sums of dH0 (fp64 on the GPU) of the same run
M = 40000
gen = torch.Generator().manual_seed(M + 1)
run = _MlpRun(M, 4, 1, dev)
run.Z.copy_((torch.randn(run.Z.shape, generator=gen) * 0.5).to(torch.bfloat16).to(dev))
xe = torch.zeros((M, 48)); xe[:, :42] = torch.randn(M, 42, generator=gen).clamp(-1, 1)
run.xenc.copy_(xe.to(dev)); run.tile_mask.fill_(31)
cc = rcfg.to_c()
_capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(pk.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), M, C.byref(run.c), st), "fwd")
dl = torch.randn(M, 4, generator=gen).to(dev)
tex = torch.full((M, 5, 4), -1, dtype=torch.int32, device=dev); tw = torch.zeros((M, 5, 4), device=dev)
for rep in range(6):
    gs = pk.grad_sink(); pk.gflat.zero_()
    dH = torch.zeros((M, 2048), dtype=torch.bfloat16, device=dev); dN = torch.zeros((3, M, 512), dtype=torch.bfloat16, device=dev)
    _capi.check(lib.scenerf_hip_mlp_backward(C.byref(cc), C.byref(pk.c), C.byref(gs), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(),
                                             tex.data_ptr(), tw.data_ptr(), M, C.byref(run.c), dl.data_ptr(), dH.data_ptr(), dN.data_ptr(), None, st), "bwd")
    torch.cuda.synchronize()
    ref = dH[:, :512].double().sum(0)
    got = pk.gviews["b_in"].double()
    err = (got - ref)
    bad = (err.abs() > 1e-4 * ref.abs().max()).nonzero().flatten().tolist()
    # is the error of a bad column the sum of one 64-row chunk (or one 384-row slice) of that column?
    expl = []
    for n in bad[:6]:
        col = dH[:, n].double()
        ch = col.view(-1, 64).sum(1)
        j = int((ch + err[n]).abs().argmin()); k = int((ch - err[n]).abs().argmin())
        expl.append("n=%d err=%.4f  (-chunk %d: %.4f | +chunk %d: %.4f)" % (n, float(err[n]), j, float(ch[j]), k, float(ch[k])))
    print("rep %d: %d bad columns %s" % (rep, len(bad), bad[:24]), expl)
