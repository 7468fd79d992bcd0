"""Does the gaussian head's forward (4,800 rows = 75 blocks of the 64-row ring kernel: a launch that leaves 70 % of the CUs idle) hide
behind half of the radiance MLP's forward (76,800 rows = 600 blocks of the 128-row kernel) when the two run on different streams?
Times: the full 153,600-row forward alone; head then full (today's order, one stream); head on a side stream beside the first half,
second half behind both (the split order).   usage: overlap_probe.py [reps]"""
import ctypes as C, dataclasses, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
from scenerf_amd.renderer import MLP_PARAM_NAMES, PackedMLP, _MlpRun

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda:0")
lib = _capi.load()
rcfg = RenderConfig.kitti(precision="bf16")
cc = rcfg.to_c()
pk = PackedMLP([torch.as_tensor(synth.mlp_state(1, 4)[n]).to(dev) for n in MLP_PARAM_NAMES], 4, rcfg)
pkg = PackedMLP([torch.as_tensor(synth.mlp_state(2, 2)[n]).to(dev) for n in MLP_PARAM_NAMES], 2, rcfg)
gen = torch.Generator().manual_seed(1)


def mk(M, d_out):
    run = _MlpRun(M, d_out, 1, dev)
    run.Z.copy_((torch.randn(run.Z.shape[0], 2480, generator=gen) * 0.5).to(torch.bfloat16))
    x = torch.randn(M, 48, generator=gen).clamp(-1, 1); x[:, 42:] = 0
    run.xenc.copy_(x); run.tile_mask.fill_(1)
    return run


full, half_a, half_b, head = mk(153600, 4), mk(76800, 4), mk(76800, 4), mk(4800, 2)
main, side = torch.cuda.current_stream(), torch.cuda.Stream()


def fwd(run, p, stream):
    _capi.check(lib.scenerf_hip_mlp_forward(C.byref(cc), C.byref(p.c), run.Z.data_ptr(), run.xenc.data_ptr(), run.tile_mask.data_ptr(), run.M,
                                            C.byref(run.c), stream.cuda_stream), "fwd")


def seq():
    fwd(head, pkg, main); fwd(full, pk, main)


def seq_halves():
    fwd(head, pkg, main); fwd(half_a, pk, main); fwd(half_b, pk, main)


def split():
    side.wait_stream(main)
    fwd(head, pkg, side)
    fwd(half_a, pk, main)
    main.wait_stream(side)
    fwd(half_b, pk, main)


def split2():   # second half on the side stream right behind the head: the two halves may overlap each other's tail rounds
    side.wait_stream(main)
    fwd(head, pkg, side)
    fwd(half_b, pk, side)
    fwd(half_a, pk, main)
    main.wait_stream(side)


for name, fn in (("full alone", lambda: fwd(full, pk, main)), ("head alone", lambda: fwd(head, pkg, main)), ("head, full (one stream: today)", seq),
                 ("head, half, half (one stream)", seq_halves), ("head || half, then half", split), ("(head, half) || half", split2)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print("%-34s %.1f us" % (name, e0.elapsed_time(e1) * 1e3 / reps), flush=True)
