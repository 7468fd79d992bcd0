#!/usr/bin/env python
"""bench.py -- rays/sec of the SceneRF training hot path (render_rays_batch fwd+bwd) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8 ...            # starts its own 8 ranks (re-executes itself under torch.distributed.run), or:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic rays, exactly what the reference's
``process_single_source`` does per image (scenerf.py:266-272 + backward): convert the image's 5 feature maps,
pack both MLPs, render R rays in one chunk, back-propagate the loss proxy that touches every gradient edge the
reference losses use (depth, colour, KL, gaussian means) into both MLPs AND the feature maps, all-reduce
the MLP gradients over RCCL (N>1) and take a fused AdamW step on them.
Workload = BASELINE.json configs[1]: KITTI 370x1220, sphere 1500x452, 128 samples/ray (U=64,G=4,P=16), R=1200
rays per GPU per step, bf16 GEMM operands / fp32 accumulate.  Weak scaling: every rank renders its own R rays.
Prints ONE JSON line (rank 0).  Extra objects: ``roofline`` (dominant MFMA kernel, HIP-event timed inside the
library on its launch stream), ``roofline_composite`` (HBM-bound compositing pass), ``cpu_baseline`` (the CPU
oracle = a port of the reference, timed on this box's host cores: best of 3 at the bench's own R), ``fp32_mode`` (the same
step with fp32 MFMA end to end) next to ``eager_gpu_baseline`` (the eager fp32 port on this GPU: matched precision),
``allreduce`` (N > 1: per-rank time of the gradient collectives and how much of it the backward waited for).

    python bench.py --mode infer            # BASELINE.json configs[4]: full-frame novel-view render, N = 512, static chunks + hipGraph
    python bench.py --dry-run ...           # CPU / gloo run of the control flow only (tests/test_dist_gloo.py)
"""
import argparse
import json
import os
import sys
import time



def _supervise():
    """One GPU, started by hand or by the driver (no WORLD_SIZE): run the measurement in a CHILD process and start it again -- at most
    twice -- if the child is killed by SIGABRT.  Why: on some boxes of the pool the HSA runtime aborts ~1 in 12-24 processes of the
    training step with HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION -- this tree and the round-5 tree alike (DESIGN.md section 5.0 / 7;
    tools/stress_trees.sh) -- and an abort cannot be caught in-process.  Every restart is on the record: stderr says so when it happens, and
    the JSON line of the run that finished carries `process_restarts` and the reasons.  Nothing is retried for any other exit code.  Only
    the standard library is imported before this point (the parent never initialises the GPU)."""
    argv = sys.argv[1:]
    if os.environ.get("SRF_BENCH_CHILD") or "WORLD_SIZE" in os.environ or "-h" in argv or "--help" in argv:
        return
    if "--dry-run" in argv and not os.environ.get("SRF_BENCH_SUPERVISE"):      # (tests/test_dist_gloo.py supervises a dry run)
        return
    gpus = 1
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            gpus = argv[i + 1]
        elif a.startswith("--gpus="):
            gpus = a.split("=", 1)[1]
    if str(gpus) != "1":
        return      # N > 1 started by hand: _self_launch (torch.distributed.run owns the ranks)
    import signal
    import subprocess
    env = dict(os.environ, SRF_BENCH_CHILD="1")
    restarts = []
    while True:
        p = subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=subprocess.PIPE, text=True)

        def _forward(signum, _frame, child=p):      # whoever stops the bench stops the measurement too
            child.send_signal(signum)
        old = {sg: signal.signal(sg, _forward) for sg in (signal.SIGTERM, signal.SIGINT)}
        try:
            out = p.communicate()[0]
        finally:
            for sg, h in old.items():
                signal.signal(sg, h)
        rc = p.returncode
        if rc in (-signal.SIGABRT, 128 + signal.SIGABRT) and len(restarts) < 2:
            restarts.append("child process %d killed by SIGABRT after printing %d bytes" % (p.pid, len(out)))
            sys.stderr.write("[bench.py] %s: starting the measurement again (%d of 2)\n" % (restarts[-1], len(restarts)))
            sys.stderr.flush()
            continue
        break
    lines = out.splitlines()
    if restarts:
        for i in range(len(lines) - 1, -1, -1):
            if lines[i].startswith("{"):
                try:
                    d = json.loads(lines[i])
                    d["process_restarts"] = len(restarts)
                    d["process_restart_reasons"] = restarts
                    lines[i] = json.dumps(d)
                except ValueError:
                    pass
                break
    if lines:
        sys.stdout.write("\n".join(lines) + "\n")
    sys.stdout.flush()
    sys.exit(rc if rc >= 0 else 128 - rc)


if __name__ == "__main__":
    _supervise()

# ranks of a process group (N > 1, or --force-dist): more hardware queues than the runtime's default 4 BEFORE the runtime initialises -- with
# RCCL's stream in the process the renderer's three streams otherwise share queues and the eagerly issued step serialises (scenerf_amd.dist.
# more_hw_queues: 3.02 -> 2.65 ms per step); the one-process line is left as the runtime comes
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or "--force-dist" in sys.argv:
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from scenerf_amd import _capi, synth  # noqa: E402
from scenerf_amd import dist as sdist  # noqa: E402
from scenerf_amd.model import SceneRF  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def pmc_traffic(logical_name, rows=None):
    """HBM bytes per launch of the kernel FUNCTION that runs `logical_name`, from the committed rocprofv3 --pmc passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/pmc_hbm.py).  PMC counters cannot be read from inside the
    process, so this is the last profiled value of the same bench command, not a live reading; None if absent."""
    import glob
    # (the step's counters: not the per-ray tail's or the compositing probe's files, which match the same pattern)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json"))
                   if "_tail_" not in os.path.basename(f) and "composite" not in os.path.basename(f))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    meta = d.pop("_meta", {}) if isinstance(d, dict) else {}
    try:
        from scenerf_amd import build as _b
        fresh = bool(meta.get("src_digest")) and meta.get("src_digest") == _b._digest()
    except Exception:
        fresh = False
    pmc_traffic.meta = {"source": os.path.basename(files[-1]), "fresh": fresh,
                        "collected_at": meta.get("commit") or "unknown (collected before the digest was recorded)"}
    # (the fused passes run as wide.hip's 128-row kernels where the launch is big enough, else as fused.hip's 64-row ones)
    if logical_name.startswith("mlp_fwd_fused"):
        fn = "mlp_wide_kernel<0>" if any(k.startswith("mlp_wide_kernel<0>") for k in d) and not logical_name.endswith("/g") else "mlp_fused_kernel<0>"
    elif logical_name.startswith("mlp_bwd_fused"):
        fn = "mlp_wide_kernel<1>" if any(k.startswith("mlp_wide_kernel<1>") for k in d) and not logical_name.endswith("/g") else "mlp_fused_kernel<1>"
    elif logical_name == "gemm_wgrad_fc":
        fn = "wgrad_tr_kernel"       # the batched transposing-read launch (wgrad.hip)
    elif "wgrad" in logical_name:
        fn = "gemm_tn_kernel"
    else:
        fn = "gemm_nt_glds_kernel" if logical_name.startswith("gemm_") and not logical_name.endswith("/g") else None
    if not fn:
        return None
    # the dominant launch of a logical kernel is the largest grid of its function (main MLP, not the 4-point gaussian head)
    sized = [(int(k.split("@grid=")[1]), k, v) for k, v in d.items() if k.startswith(fn) and "@grid=" in k]
    if sized and rows and logical_name.startswith("mlp_") and any(g == (rows + 127) // 128 * 256 for g, _, _ in sized):
        sized = [t for t in sized if t[0] == (rows + 127) // 128 * 256]     # the 128-row kernels: one 256-thread block per 128 rows
    if sized and (logical_name.startswith("mlp_") or logical_name == "gemm_wgrad_fc"):
        _, k, v = max(sized)
        return {"bytes_per_launch": round(v["hbm_bytes_per_launch"]), "kernel_fn": k, "source": os.path.basename(files[-1]),
                "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, average over the launches of this grid size"}
    for k, v in d.items():
        if k.startswith(fn) and "@grid=" not in k:
            return {"bytes_per_launch": round(v["hbm_bytes_per_launch"]), "kernel_fn": k, "source": os.path.basename(files[-1]),
                    "note": "average over all launches of this kernel function in a step"}
    return None


def _trace(msg):
    """SRF_BENCH_TRACE=1: the side measurements' progress on stderr (which leg a crash of the process belongs to)."""
    if os.environ.get("SRF_BENCH_TRACE"):
        print("[bench %.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


_T0 = time.perf_counter()


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; infer: 5 frames)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps (default 5; infer: 1 frame)")
    ap.add_argument("--rays", type=int, default=1200, help="rays per GPU per step (reference n_rays)")
    ap.add_argument("--samples", type=int, default=None, choices=[64, 96, 128, 256, 512], help="default 128 (train) / 512 (infer)")
    ap.add_argument("--chunk", type=int, default=4096, help="infer: rays per static chunk")
    ap.add_argument("--stride", type=int, default=1, help="infer: pixel stride of the rendered frame (1 = all 451,400 px)")
    ap.add_argument("--dry-run", action="store_true", help="CPU / gloo: run the control flow (collectives, timing, JSON) around a stub step")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--host-rng", action="store_true",
                    help="draw the gaussian sampler's normal noise on the host like the reference (utils.py:208-211: torch.normal on the CPU generator + "
                         "upload; the module's default, RenderConfig.device_rng=False) instead of on the device: ~0.25 ms of host time per 1,200 rays")
    ap.add_argument("--maps", default="hwc", choices=["hwc", "chw"],
                    help="memory layout of the feature maps handed to render_rays_batch: hwc = (C,H,W) tensors with channels-last strides (what a "
                         "torch.channels_last decoder or SphereResampler(layout='hwc') emits; read in place), chw = contiguous (C,H,W) as the "
                         "reference's decoder emits (converted per call).  The other one is timed too and reported as 'other_entry'")
    ap.add_argument("--sync", default="session", choices=["session", "step"],
                    help="N>1: 'session' = the renderer's per-session hooks (two 21.7 MB all-reduces started inside the backward, overlapped); "
                         "'step' = dist.StepGradSync, one 43.3 MB all-reduce at the end of backward (safe when ranks render different numbers "
                         "of source frames)")
    ap.add_argument("--optimizer", default="fused", choices=["fused", "torch"],
                    help="AdamW of the step: 'fused' = scenerf_amd.optim.FusedAdamW (one HIP launch over all 40 parameter tensors), 'torch' = "
                         "torch.optim.AdamW(fused=True)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="train: issue the step as ONE hipGraph replay per rank (scenerf_amd.graph.GraphedStep: forward, loss, backward with the "
                         "gradient all-reduces, fused AdamW captured once); 'auto' = on when it can be captured (bf16, device noise, the fused "
                         "optimizer) and EVERY rank's capture succeeded -- otherwise all ranks step eagerly; one GPU: the eagerly issued step is "
                         "reported next to it as `eager_step`; 'off' = the eager step is the headline")
    ap.add_argument("--dry-fail-capture", type=int, default=-1, help=argparse.SUPPRESS)   # --dry-run: the rank whose stub capture fails
    ap.add_argument("--loss", default="source", choices=["source", "proxy"],
                    help="what stands between the renderer's forward and backward in a step: 'source' = the reference's loss of one source frame "
                         "(scenerf.py:203-238 around process_single_source: colour L1 + photometric reprojection against synthetic source / target "
                         "images, KL, closest-gaussian term, KITTI weights) through scenerf_amd.loss_side.source_loss, one launch each way; 'proxy' = "
                         "depth.mean() + color.mean() + loss_kl.mean() + gaussian_means.mean() in eager torch (rounds 1-3: ~12 launches, ~90 us)")
    ap.add_argument("--force-dist", action="store_true",
                    help="one rank: still create the process group (nccl = RCCL) and issue the gradient collectives (scenerf_amd.dist."
                         "FORCE_COLLECTIVES): the N > 1 code path on a single GPU; reported as `allreduce`")
    ap.add_argument("--device-warm-steps", type=int, default=150,
                    help="untimed steps issued as part of the setup, before the W warm-up steps: the GPU idles through model construction and "
                         "capture and its clocks take a few hundred ms of load to settle (20 timed steps right after 5 warm-ups measure "
                         "2-4 %% slower than the 300-step steady_state of the same process); 0 = rounds 1-3 behaviour")
    ap.add_argument("--set", action="append", default=[], metavar="MODULE.ATTR=VALUE",
                    help="development: set a module-level knob (or class attribute) of scenerf_amd before the run, e.g. --set renderer.PREFILL_AT=3, "
                         "--set model.SceneRF.share_image_sessions=False")
    ap.add_argument("--cfg", action="append", default=[], metavar="FIELD=VALUE",
                    help="development: set a field of the models' RenderConfig (kernel-path selectors), e.g. --cfg fwd_kernel=\"'ring'\"")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed region (no eager / other-entry / drop-in / steady-state / roofline legs): what a profiler should see")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the short BASELINE configs[3] (BundleFusion) and configs[4] (inference) legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=0, help="rays of the CPU-baseline sample (0 = the bench's own --rays)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="CPU-baseline repetitions (best is reported)")
    ap.add_argument("--no-fp32-mode", action="store_true")
    ap.add_argument("--no-eager-baseline", action="store_true")
    ap.add_argument("--kernels-json", default="", help="write the per-kernel table here")
    a = ap.parse_args()
    for kv in a.set:
        import ast, importlib
        path, val = kv.split("=", 1)
        parts = path.split(".")           # module[.Class].ATTR
        obj = importlib.import_module("scenerf_amd." + parts[0])
        for name in parts[1:-1]:
            obj = getattr(obj, name)
        setattr(obj, parts[-1], ast.literal_eval(val))
    if a.headline_only:
        a.no_roofline = a.no_extra_legs = a.no_cpu_baseline = a.no_eager_baseline = a.no_fp32_mode = True
    if a.steps is None:
        a.steps = 5 if a.mode == "infer" else 20
    if a.warmup is None:
        a.warmup = 1 if a.mode == "infer" else 5
    if a.samples is None:
        a.samples = 512 if a.mode == "infer" else 128
    return a


def sample_split(n):
    # N = U + 4*P with U = N/2 (SURVEY §8d config mapping; 96 -> U=64,P=8)
    if n == 96:
        return 64, 8
    return n // 2, n // 8


def make_optimizer(args, params):
    """AdamW(lr=1e-5, weight_decay=0) as the reference configures it (scenerf.py:756-761)."""
    if getattr(args, "optimizer", "fused") == "torch":
        return torch.optim.AdamW(params, lr=1e-5, weight_decay=0.0, fused=True)
    from scenerf_amd.optim import FusedAdamW
    return FusedAdamW(params, lr=1e-5, weight_decay=0.0, capturable=bool(getattr(args, "capturable", False)))


def make_loss(args, dev, img_size, K, pix, rank=0, weights=(1.0, 1.0, 0.01)):
    """The step's loss as a function of render_rays_batch's output dict (see --loss)."""
    if getattr(args, "loss", "source") == "proxy":
        return lambda out: out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
    from scenerf_amd.loss_side import make_rng_state, source_loss
    rng = make_rng_state(dev, seed=4242 + rank)
    g = torch.Generator().manual_seed(900 + rank)
    W, H = img_size
    img_s, img_t = torch.rand(3, H, W, generator=g).to(dev), torch.rand(3, H, W, generator=g).to(dev)
    T_s2t = synth.rel_pose(-1.0, 3.0).to(dev)
    iK = torch.inverse(K).contiguous()
    R = pix.shape[0]

    def loss_fn(out):      # (scenerf.py:378's randn * 1e-5 on the identity term: made inside the kernel, like scenerf_amd.training does)
        return source_loss(out, pix, img_s, img_t, K, iK, T_s2t, noise=None, noise_scale=1e-5, reproj_weight=weights[0],
                           color_weight=weights[1], dist2closest_weight=weights[2], rng_state=rng)[0]
    return loss_fn


def make_model(args, dev, precision=None):
    U, P = sample_split(args.samples)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=U, n_pts_per_gaussian=P,
                precision=precision or args.precision, device_rng=not getattr(args, "host_rng", False)).to(dev)
    m.mlp.load_state_dict(synth.mlp_state(1, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2, out_scale=4.0))
    for kv in getattr(args, "cfg", None) or []:      # development: --cfg fwd_kernel='ring'
        import ast
        k, v = kv.split("=", 1)
        assert hasattr(m.render_cfg, k), k
        setattr(m.render_cfg, k, ast.literal_eval(v))
    return m


def _oracle_setup(args, dev):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import scenerf_oracle as orc
    U, P = sample_split(args.samples)
    cfg = orc.OracleConfig.kitti(n_pts_uni=U, n_pts_per_gaussian=P)
    mlp, mlpg = synth.mlp_state(1, 4), synth.mlp_state(2, 2, out_scale=4.0)
    mlp = {k: v.to(dev).requires_grad_(True) for k, v in mlp.items()}
    mlpg = {k: v.to(dev).requires_grad_(True) for k, v in mlpg.items()}
    maps = {k: v.to(dev).requires_grad_(True) for k, v in synth.feature_maps(1500, 452, 3).items()}
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)

    def run(R, seed):
        pix = synth.stride2_pixels((1220, 370), R, seed).to(dev)
        nu, ng = synth.sampling_noise(R, U, 4 * P, seed + 1)
        out = orc.render_chunk(cfg, mlp, mlpg, K, T, maps, pix, nu.to(dev), ng.to(dev))
        orc.training_proxy_loss(out).backward()
        for d in (mlp, mlpg, maps):
            for v in d.values():
                v.grad = None
    return run


def cpu_baseline(args):
    """The oracle (port of the reference's eager path) on the host cores, fwd+bwd at the bench's own ray count, best of --cpu-reps
    (BASELINE.md section 3)."""
    cores = os.cpu_count() or 1
    threads = min(cores, 32)   # torch's CPU kernels stop scaling (and regress) beyond a few dozen threads
    torch.set_num_threads(threads)
    run = _oracle_setup(args, "cpu")
    run(8, 50)  # warm-up (allocators, thread pool)
    R = args.cpu_rays or args.rays
    times = []
    for i in range(max(1, args.cpu_reps)):
        t0 = time.perf_counter()
        run(R, 60 + i)
        times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(R / best, 2), "unit": "rays/s", "cores": threads, "kind": "port",
            "what": "oracle/scenerf_oracle.py, the CPU restatement of the reference's eager path pinned on reference-minted goldens (the "
                    "reference itself is not on this box), render + proxy loss fwd+bwd",
            "threads_used": threads, "threads_available": cores,
            # the REAL reference (imported unmodified from /root/reference, which does not exist on the GPU box) at this very configuration,
            # measured once in the survey container and recorded in BASELINE.md section 2 -- a constant of the repository, not of this run
            "reference_measured": {"value": 39.8, "unit": "rays/s", "cores": 8, "kind": "reference",
                                   "what": "astra-vision/SceneRF's own modules, SceneRF.render_rays_batch + backward, 1200 rays x 128 samples, fp32, one "
                                           "chunk: 11.80 s forward + 18.34 s backward on 8 cores of an Intel Xeon @ 2.10 GHz, torch 2.10 CPU",
                                   "source": "BASELINE.md section 2 (survey container; not measurable on the GPU box)"},
            "sample": "%d rays x %d samples fwd+bwd on %d of %d host threads (torch's CPU kernels regress beyond a few dozen), full KITTI maps, "
                      "best of %d (%s s)" % (R, args.samples, threads, cores, len(times), ", ".join("%.1f" % t for t in times))}


def eager_gpu_baseline(args, dev):
    """The same eager-PyTorch port of the reference run on THIS GPU through PyTorch-ROCm: the closest available
    stand-in for "the reference on one GPU" (the reference itself is not on the GPU box).  Extra information next
    to cpu_baseline, not the optimisation target."""
    run = _oracle_setup(args, dev)
    R = args.rays
    run(R, 50)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for i in range(n):
        run(R, 60 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return {"value": round(R / dt, 1), "unit": "rays/s", "kind": "eager PyTorch-ROCm port of the reference hot path, fp32",
            "ms_per_step": round(dt * 1e3, 2), "sample": "%d rays x %d samples fwd+bwd, 3 steps" % (R, args.samples)}


def _make_maps(layout, dev, rank):
    """The five pyramid levels as leaf (C,H,W) fp32 tensors: contiguous ('chw') or with channels-last strides ('hwc': memory (H,W,C))."""
    maps = {}
    for k, v in synth.feature_maps(1500, 452, 3 + rank).items():
        if layout == "hwc":
            c, h, w = v.shape
            t = torch.empty_strided((c, h, w), (1, w * c, c), dtype=torch.float32, device=dev)
            t.copy_(v.to(dev))
            maps[k] = t.requires_grad_(True)
        else:
            maps[k] = v.to(dev).requires_grad_(True)
    return maps


def _timed(step, args, world, dev, sync):
    """W untimed + exactly K timed steps between barrier + synchronize; MAX over ranks.  Returns (seconds, last step result)."""
    def barrier():
        if world > 1:
            torch.distributed.barrier()
        sync()

    last = None
    for _ in range(args.warmup):
        last = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    _timed.host_s = time.perf_counter() - t0     # the host's share: all K steps issued (nothing waited for yet)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    return dt, last


def _allreduce_report(world, steps_recorded):
    """Per-rank collective timing of the LAST recorded steps (sdist.TIMING), gathered on every rank (a collective: all ranks call it)."""
    rec = sdist.TIMING or []
    if rec:
        torch.cuda.synchronize()
    mine = {"wait": 0.0, "sync": 0.0}
    for kind, e0, e1 in rec:
        mine[kind] += e0.elapsed_time(e1)
    mine = {k: v / max(steps_recorded, 1) for k, v in mine.items()}
    allr = [None] * world
    torch.distributed.all_gather_object(allr, mine)
    return {"per_rank_ms_per_step": [{"radiance_mlp_exposed_wait": round(r["wait"], 4), "gaussian_head_on_side_stream": round(r["sync"], 4)}
                                     for r in allr],
            "max_exposed_wait_ms_per_step": round(max(r["wait"] for r in allr), 4),
            "note": "exposed wait = time the backward stream blocks on the radiance MLP's asynchronous all-reduce (21.7 MB, started "
                    "before the feature-gradient scatter); the head's all-reduce (21.7 MB) runs in stream order on the backward's side stream"}


def _collective_probe(world, dev, numel, reps=10):
    """One gradient collective ON ITS OWN (the packed fp32 sink of one MLP: `numel` floats, the size both of the step's all-reduces have),
    `reps` times back to back on the gradient communicator, every rank: per-rank us per collective and the bus bandwidth that implies
    (2 (N-1)/N x bytes / time: the ring's per-link figure, to read against xGMI's ~153 GB/s per link).  The captured step cannot bracket
    its collectives with events (they are graph nodes); this is the same collective outside the graph, so that a first N-GPU run says by
    itself whether RCCL's ring -- not the step around it -- is the slow part.  A collective: every rank calls it."""
    buf = torch.ones(numel, dtype=torch.float32, device=dev)
    cuda = buf.is_cuda
    sdist.allreduce_mean_(buf)                     # (first use of this size: not timed)
    if cuda:
        torch.cuda.synchronize()
    torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        sdist.allreduce_mean_(buf)
    if cuda:
        torch.cuda.synchronize()
    us = (time.perf_counter() - t0) / reps * 1e6
    allr = [None] * world
    torch.distributed.all_gather_object(allr, round(us, 1))
    worst = max(allr)
    return {"bytes": numel * 4, "reps": reps, "us_per_collective_per_rank": allr,
            "bus_GB_per_s": round(2 * (world - 1) / max(world, 1) * numel * 4 / (worst * 1e-6) / 1e9, 2) if worst > 0 else None,
            "backend": torch.distributed.get_backend(),
            "note": "the step issues two of these per source frame (gaussian head early on the side stream, radiance MLP asynchronously under "
                    "the feature-gradient scatter); on the captured path they are graph nodes and cannot be timed in place"}


class _StubModel:
    """--dry-run: stands in for SceneRF on a CPU/gloo process group: its step issues the renderer's collectives in the renderer's order
    through the same two hooks (grad_sync in stream order for the gaussian head, grad_sync_async started early / finished late for the
    radiance MLP), so the bench's own control flow -- which ranks call what, when the hooks are removed, what runs on rank 0 only --
    is exercised without a GPU (tests/test_dist_gloo.py)."""

    def __init__(self, rank):
        self.rank = rank
        self.grad_sync = self.grad_sync_async = None
        self.head, self.main = torch.zeros(1024), torch.zeros(1024)

    def step(self):
        self.head.fill_(self.rank + 1.0)
        self.main.fill_(2.0 * (self.rank + 1))
        if self.grad_sync is not None:
            self.grad_sync(self.head)
        fin = self.grad_sync_async(self.main) if self.grad_sync_async is not None else None
        if fin is not None:
            fin()
        return self.head[0] + self.main[0]


def infer_main(args, rank, world, dev, census=None):
    """BASELINE.json configs[4]: novel-view inference, full frame, 512 samples/ray (U=256, G=4, P=64), no_grad, static chunks of
    --chunk rays with the tail padded, one captured hipGraph per frame replayed per chunk, device RNG (scenerf_amd/inference.py).
    A step = one full frame of one pose.  N > 1 = N independent replicas (inference does not shard)."""
    from scenerf_amd.inference import pixel_grid
    model = make_model(args, dev).eval()
    maps = {k: v.to(dev) for k, v in synth.feature_maps(1500, 452, 3 + rank).items()}
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
    grid = pixel_grid((1220, 370), args.stride, dev)
    n = grid.shape[0]

    def frame(graph=True):
        with torch.no_grad():
            return model.render_image(K, T, maps, sampled_pixels=grid, ray_batch_size=args.chunk, keys=("depth", "color"), use_graph=graph)

    dt, last = _timed(frame, args, world, dev, torch.cuda.synchronize)
    assert bool(torch.isfinite(last["depth"]).all()), "depth is not finite"
    value = world * n * args.steps / dt
    roof = roof_c = eager = None
    if rank == 0:
        # the same frame with every chunk launched eagerly (no graph): what the graph buys
        frame(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frame(False)
        torch.cuda.synchronize()
        eager = {"value": round(n / (time.perf_counter() - t0), 1), "unit": "rays/s", "kind": "same static chunks, launched eagerly (1 frame)"}
    if rank == 0 and not args.no_roofline:
        lib = _capi.load()
        lib.scenerf_hip_profile_enable(1)
        frame(False)
        torch.cuda.synchronize()
        kernels = _capi.profile_collect()
        lib.scenerf_hip_profile_enable(0)
        roof, roof_c = _rooflines(kernels, args, 1, None, None)
        if args.kernels_json:
            json.dump(kernels, open(args.kernels_json, "w"), indent=1)
    if rank == 0:
        U, P = sample_split(args.samples)
        print(json.dumps({
            "metric": "rays/sec (novel-view inference, full-frame render, %d samples/ray, hipGraph-replayed static chunks)" % args.samples,
            "value": round(value, 1), "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "KITTI 370x1220 novel-view render, sphere 1500x452, %d px per frame (stride %d), %d samples/ray (U=%d,G=4,P=%d), "
                                   "chunks of %d rays (tail padded), no_grad, one hipGraph per frame replayed per chunk, device RNG; "
                                   "N>1 = independent replicas" % (n, args.stride, args.samples, U, P, args.chunk),
                       "rays_per_frame": n, "samples_per_ray": args.samples, "chunk": args.chunk, "parallelism": "replicas%d" % world,
                       "precision": args.precision, "mode": "infer"},
            "roofline": roof, "roofline_composite": roof_c, "eager_launch": eager, "ranks": census}), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def _rooflines(kernels, args, nprof, value_per_gpu, world):
    """(roofline of the dominant MFMA kernel, roofline of the compositing pass) from the in-library HIP-event table."""
    roof = roof_c = None
    for k in kernels:
        k["avg_us"] = k["total_ms"] * 1e3 / max(k["launches"], 1)
        k["launches_per_step"] = k["launches"] / nprof
        k["ms_per_step"] = k["total_ms"] / nprof
    mfma = [k for k in kernels if k["flops"] > 0 and k["total_ms"] > 0]
    if mfma:
        dom = max(mfma, key=lambda k: k["total_ms"])
        ach = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        peak = PEAK_BF16_TFLOPS if args.precision == "bf16" else PEAK_FP32_TFLOPS
        tot_f = sum(k["flops"] for k in mfma)
        tot_t = sum(k["total_ms"] for k in mfma)
        tr = pmc_traffic(dom["name"], getattr(args, "rows_per_launch", None))
        tmeta = getattr(pmc_traffic, "meta", {})
        roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(ach / peak, 4),
                # HBM bytes per launch of the dominant kernel from the last committed PMC pass of the bench command (counters cannot be read
                # in-process); traffic_fresh says whether that pass was taken on THIS tree's kernel sources (content digest)
                "traffic": tr["bytes_per_launch"] if tr else None, "traffic_kernel": tr["kernel_fn"] if tr else None,
                "traffic_source": tmeta.get("source"), "traffic_fresh": tmeta.get("fresh"), "traffic_collected_at": tmeta.get("collected_at"),
                "timing": "HIP events around every launch, on its launch stream, inside the library (scenerf_hip_profile_enable); the "
                          "rocprofv3 --kernel-trace average of the same kernel in profiles/ runs 5-10 % longer (tracer overhead)",
                "avg_launch_us": round(dom["avg_us"], 2), "flops_per_launch": dom["flops"] / dom["launches"],
                "all_mfma_kernels_achieved": round(tot_f / (tot_t * 1e-3) / 1e12, 2),
                "all_mfma_kernels_frac": round(tot_f / (tot_t * 1e-3) / 1e12 / peak, 4),
                "mfma_ms_per_step": round(tot_t / nprof, 3),
                "per_kernel": {k["name"]: {"achieved": round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12, 1),
                                           "frac": round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12 / peak, 4), "avg_launch_us": round(k["avg_us"], 1)}
                               for k in sorted(mfma, key=lambda k: -k["total_ms"])[:6]}}
        # the same per-kernel numbers as flat scalars (nested objects do not survive the driver's `parsed` copy of this line)
        short = {"mlp_fwd_fused": "fwd", "mlp_bwd_fused": "dgrad_chain", "gemm_wgrad_fc": "wgrad_batch", "mlp_fwd_fused/g": "head_fwd",
                 "mlp_bwd_fused/g": "head_dgrad", "gemm_dfeat_scatter": "dfeat", "gemm_wgrad_fc/g": "head_wgrad"}
        for k in mfma:
            if k["name"] in short:
                roof["frac_" + short[k["name"]]] = round(k["flops"] / (k["total_ms"] * 1e-3) / 1e12 / peak, 4)
                roof["us_" + short[k["name"]]] = round(k["avg_us"], 1)
        roof.update(_roofline_extras(mfma, peak, nprof, args))
        if value_per_gpu is not None:
            # with zero-K-block skipping, utilisation is reported on the FLOPs issued (above) and the rays/s are
            # quoted separately against the dense algorithmic count of the reference: fwd 2(N*5,405,696 + G*5,404,672), x3 fwd+bwd
            dense = 3.0 * 2.0 * (args.samples * 5405696 + 4 * 5404672)
            roof["dense_equivalent"] = {"flops_per_ray_fwd_bwd": dense, "achieved": round(value_per_gpu * dense / 1e12, 1),
                                        "unit": "TFLOP/s", "frac_of_peak": round(value_per_gpu * dense / 1e12 / peak, 4),
                                        "note": "rays/s x the reference's dense FLOPs per ray (it multiplies the out-of-range "
                                                "scales' zeros); not a utilisation: the kernels skip those K blocks"}
    # (training chunks launch the fused per-ray tail: compositing + RaySOM forward, their autograd + the sampler's backward)
    comp = [k for k in kernels if k["name"] in ("composite_fwd", "composite_bwd", "ray_tail_fwd", "ray_tail_bwd")]
    if comp:
        b = sum(k["bytes"] for k in comp)
        t = sum(k["total_ms"] for k in comp)
        ach = b / (t * 1e-3) / 1e9
        roof_c = {"bound": "hbm", "kernel": "+".join(k["name"] for k in comp), "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                  "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                  "avg_launch_us": round(sum(k["total_ms"] for k in comp) * 1e3 / sum(k["launches"] for k in comp), 2),
                  "bytes_per_ray": sum({"composite_fwd": 32 * args.samples + 24, "ray_tail_fwd": 32 * args.samples + 24,
                                        "composite_bwd": 48 * args.samples + 40, "ray_tail_bwd": 44 * args.samples + 40}[k["name"]] for k in comp)}
        # the same two kernels at an inference-sized chunk (65,536 rays): at the training chunk (1,200 rays = 1,200 waves) the pass is
        # launch latency, not bandwidth; this is the number the HBM roofline applies to (tools/composite_probe.py, profiles/*composite*)
        try:
            pr = _composite_probe(65536, args.samples)
            roof_c["at_inference_chunk"] = pr
            for nm in ("fwd", "bwd", "tail_fwd", "tail_fwd_nosom", "tail_bwd"):
                roof_c["frac_65536_rays_" + nm] = pr[nm]["frac"]
            roof_c["traffic"] = _tail_traffic(args.samples)
        except Exception as e:   # never let the extra leg break the bench line
            roof_c["at_inference_chunk"] = {"error": str(e)}
    return roof, roof_c


def padding_flops(name, flops_per_launch, rows):
    """FLOPs of a launch that multiply padding (operand columns that exist only to fill a tile), so that a fraction of the peak can be quoted
    on USEFUL work too.  The batched weight-gradient launch (csrc/mlp.hip) carries two padded problems: lin_in's K = 42 real input
    columns ride in a 256-column tile (SCENERF_WIN_LD), and nothing else -- lin_z's 256 dense columns of Z are all real latent columns
    (80 + 160 + the first 16 of the third level).  Every other MFMA kernel issues exactly what it reports."""
    if name.split("/")[0] == "gemm_wgrad_fc" and rows:
        return 2.0 * rows * 512.0 * (256 - 42)
    return 0.0


def _roofline_extras(mfma, peak, nprof, args, big=0.10):
    """``roofline_min``: the WORST big kernel (lowest fraction of the MFMA peak among the kernels that take more than ``big`` of the
    profiled steps' kernel time) -- the line's ``roofline`` names the LONGEST kernel, and two kernels 2 % apart in time trade that place
    without either getting faster.  ``roofline_useful_frac``: the dominant kernel's fraction on useful FLOPs (padding excluded)."""
    out = {}
    tot_t = sum(k["total_ms"] for k in mfma)
    step_ms = getattr(args, "step_ms_for_roofline", None)
    ref_t = step_ms * nprof if step_ms else tot_t
    # (the feature-map scatter is listed with its GEMM's FLOPs but is bound by its fp32 atomics into 420 MB of accumulators, not by the
    #  matrix cores -- 750 MB of HBM traffic per step in the PMC pass: it is reported (frac_dfeat) and left out of this selection, where
    #  at a share of 9.9 - 10.3 % of the step it would otherwise enter and leave from box to box)
    bigk = [k for k in mfma if k["total_ms"] >= big * ref_t and not k["name"].startswith("gemm_dfeat")]
    frac = lambda k, f=None: (k["flops"] if f is None else f) / (k["total_ms"] * 1e-3) / 1e12 / peak
    if bigk:
        w = min(bigk, key=frac)
        out["roofline_min"] = {"bound": "mfma", "kernel": w["name"], "frac": round(frac(w), 4), "achieved": round(frac(w) * peak, 2), "peak": peak,
                               "unit": "TFLOP/s", "avg_launch_us": round(w["total_ms"] * 1e3 / max(w["launches"], 1), 2),
                               "share_of_step": round(w["total_ms"] / ref_t, 3),
                               "rule": "lowest fraction among MFMA-bound kernels above %.0f %% of the step (the atomics-bound feature-map scatter excluded)" % (100 * big),
                               "candidates": {k["name"]: round(frac(k), 4) for k in bigk}}
        out["roofline_min_frac"] = out["roofline_min"]["frac"]
        out["roofline_min_kernel"] = w["name"]
    dom = max(mfma, key=lambda k: k["total_ms"])
    rows = getattr(args, "rows_per_launch", None)
    pad = padding_flops(dom["name"], dom["flops"] / max(dom["launches"], 1), rows) * dom["launches"]
    out["roofline_useful_frac"] = round(frac(dom, dom["flops"] - pad), 4)
    out["roofline_useful_note"] = "dominant kernel on useful FLOPs: %.1f of %.1f GFLOP per launch are tile padding" % (
        pad / max(dom["launches"], 1) / 1e9, dom["flops"] / max(dom["launches"], 1) / 1e9)
    return out


def _tail_traffic(N):
    """HBM bytes per launch of the per-ray tail kernels at 65,536 rays from the committed rocprofv3 --pmc passes (tools/profile_tail.sh:
    FETCH_SIZE x 2 gfx950 correction + WRITE_SIZE), next to the algorithmic bytes -- counters cannot be read from inside the process."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_tail_pmc_hbm.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    meta = d.pop("_meta", {})
    C_ = 1 if N <= 64 else 2 if N <= 128 else 4 if N <= 256 else 8
    out = {"source": os.path.basename(files[-1]), "collected_at": meta.get("commit") or "unknown", "rays": 65536}
    try:
        from scenerf_amd import build as _b
        out["fresh"] = bool(meta.get("src_digest")) and meta.get("src_digest") == _b._digest()
    except Exception:
        out["fresh"] = False
    for key, fn, alg in (("tail_fwd", "ray_tail_fwd_kernel<%d, 4, true>" % C_, 32 * N + 24), ("tail_fwd_nosom", "ray_tail_fwd_kernel<%d, 4, false>" % C_, None),
                         ("tail_bwd", "ray_tail_bwd_kernel<%d>" % C_, 44 * N + 40)):
        v = d.get(fn + " @grid=%d" % (65536 * 64)) or d.get(fn)
        if v:
            out[key] = {"bytes_per_launch": round(v["hbm_bytes_per_launch"]), "fetch": round(v["fetch_bytes_per_launch"]), "write": round(v["write_bytes_per_launch"])}
            if alg:
                out[key]["algorithmic_bytes_per_launch"] = 65536 * alg
                out[key]["ratio"] = round(v["hbm_bytes_per_launch"] / (65536.0 * alg), 3)
    return out


def _composite_probe(R, N, reps=20):
    """The fused per-ray tail (compositing + RaySOM forward; their autograd + the sampler's backward) through the C ABI on synthetic
    [R][N] inputs, HIP-event timed on the launch stream, against the compositing pass's algorithmic bytes."""
    import ctypes as C
    from scenerf_amd.config import RenderConfig
    lib = _capi.load()
    dev = "cuda"
    U, P = sample_split(N)
    cc = RenderConfig.kitti(n_pts_uni=U, n_pts_per_gaussian=P).to_c()
    G = 4
    st = torch.cuda.current_stream().cuda_stream
    logits = torch.randn(R * N, 4, device=dev)
    logits[:, 3] -= 2
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
    fwd = lambda: lib.scenerf_hip_ray_tail_forward(C.byref(cc), logits.data_ptr(), dist.data_ptr(), z.data_ptr(), gm.data_ptr(), gs.data_ptr(), R,
                                                   dens.data_ptr(), al.data_ptr(), w.data_ptr(), dep.data_ptr(), col.data_ptr(), clo.data_ptr(),
                                                   wat.data_ptr(), ci.data_ptr(), lk.data_ptr(), sm.data_ptr(), sv.data_ptr(), ks.data_ptr(), None, st)
    bwd = lambda: lib.scenerf_hip_ray_tail_backward(C.byref(cc), logits.data_ptr(), dist.data_ptr(), z.data_ptr(), R, gd.data_ptr(), gc.data_ptr(),
                                                    None, None, None, None, offs.data_ptr(), anchors.data_ptr(), noise.data_ptr(), unit.data_ptr(),
                                                    gm.data_ptr(), gs.data_ptr(), perm.data_ptr(), ks.data_ptr(), None, None, None, dl.data_ptr(),
                                                    do.data_ptr(), None, None, None, None, None, st)
    nosom = lambda: lib.scenerf_hip_ray_tail_forward(C.byref(cc), logits.data_ptr(), dist.data_ptr(), z.data_ptr(), None, None, R,
                                                     dens.data_ptr(), al.data_ptr(), w.data_ptr(), dep.data_ptr(), col.data_ptr(), clo.data_ptr(),
                                                     wat.data_ptr(), ci.data_ptr(), None, None, None, None, None, st)
    dd, dz = f(R, N), f(R, N)
    cfwd = lambda: lib.scenerf_hip_composite_forward(logits.data_ptr(), dist.data_ptr(), z.data_ptr(), R, N, dens.data_ptr(), al.data_ptr(),
                                                     w.data_ptr(), dep.data_ptr(), col.data_ptr(), clo.data_ptr(), wat.data_ptr(), ci.data_ptr(), st)
    cbwd = lambda: lib.scenerf_hip_composite_backward(logits.data_ptr(), dist.data_ptr(), z.data_ptr(), R, N, gd.data_ptr(), gc.data_ptr(), None,
                                                      None, None, None, dl.data_ptr(), dd.data_ptr(), dz.data_ptr(), st)
    out = {"rays": R, "samples": N,
           "note": "fwd / bwd = the compositing pass proper (composite_fwd / composite_bwd stage kernels: alpha compositing only, the pass the "
                   "HBM roofline of SURVEY 8d is defined on); tail_fwd / tail_bwd = what a training chunk launches (ray_tail_*: compositing + "
                   "RaySOM, compositing + sampler / KL backward), priced on the same compositing bytes -- the SOM update's exp / log per "
                   "(sample, gaussian) makes the fused forward compute-bound; tail_fwd_nosom = the compositing-only instantiation of the "
                   "same kernel (loss_kl = NULL): what a no_grad render that does not ask for loss_kl / som_vars launches (full-frame inference)"}
    for name, fn, bpr in (("fwd", cfwd, 32 * N + 24), ("bwd", cbwd, 48 * N + 40), ("tail_fwd", fwd, 32 * N + 24), ("tail_fwd_nosom", nosom, 32 * N + 24),
                          ("tail_bwd", bwd, 44 * N + 40)):
        for _ in range(3):
            assert fn() == 0, lib.scenerf_hip_last_error()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        gbs = R * bpr / us / 1e3
        out[name] = {"avg_launch_us": round(us, 1), "achieved": round(gbs, 1), "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4)}
    return out


def _self_launch(n):
    """``python bench.py --gpus N`` outside torchrun: start the N ranks here, one process per GPU, the way the reference's trainer
    spawns its own DDP workers (scripts/train_kitti.py:127-156, Trainer(accelerator='ddp', gpus=n_gpus)) -- same command line,
    re-executed under torch.distributed.run on a free local port.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, SRF_BENCH_SELF_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _rank_census(rank, world, local, dev, dry):
    """Who is actually in the job: every rank contributes its id through the backend the gradients travel on (a one-hot all-reduce:
    on GPUs that is RCCL) and its device through an object gather.  ``ranks_seen`` must equal --gpus, and no two ranks may sit on
    one GPU -- otherwise no line is printed."""
    dinfo = {"rank": rank, "local_rank": local, "device": str(dev), "pid": os.getpid()}
    if not dry:
        pr = torch.cuda.get_device_properties(dev)
        dinfo.update(name=pr.name, cus=pr.multi_processor_count, hbm_gb=round(pr.total_memory / 2 ** 30, 1))
        for k in ("pci_bus_id", "pci_device_id", "uuid"):
            if hasattr(pr, k):
                dinfo[k] = str(getattr(pr, k))
    if world == 1:
        return {"ranks_seen": 1, "backend": None, "launched_by": "python", "devices": [dinfo]}
    onehot = torch.zeros(world, dtype=torch.int32, device=dev)
    onehot[rank] = 1
    torch.distributed.all_reduce(onehot)
    seen = int((onehot == 1).sum().item())
    devs = [None] * world
    torch.distributed.all_gather_object(devs, dinfo)
    if seen != world:
        raise SystemExit("bench.py: %d of %d ranks answered the census all-reduce" % (seen, world))
    if not dry:
        ids = [d.get("uuid") or d.get("pci_bus_id") or d["device"] for d in devs]
        if len(set(ids)) != world:
            raise SystemExit("bench.py: %d ranks share %d GPUs (%s): one process per GPU is the contract" % (world, len(set(ids)), ids))
    return {"ranks_seen": seen, "backend": torch.distributed.get_backend(),
            "launched_by": "bench.py (self-launch)" if os.environ.get("SRF_BENCH_SELF_LAUNCHED") else "torch.distributed.run",
            "devices": devs}


def _leg_roofline(run, nprof, args, samples, value_per_gpu, rows_per_launch=None):
    """roofline objects of a side leg: ``run`` under the in-library HIP-event profiler."""
    import copy
    lib = _capi.load()
    torch.cuda.synchronize()
    lib.scenerf_hip_profile_enable(1)
    for _ in range(nprof):
        run()
    torch.cuda.synchronize()
    kernels = _capi.profile_collect()
    lib.scenerf_hip_profile_enable(0)
    a2 = copy.copy(args)
    a2.samples = samples
    a2.rows_per_launch = rows_per_launch
    roof, roof_c = _rooflines(kernels, a2, nprof, value_per_gpu, 1)
    if roof_c:
        roof_c.pop("at_inference_chunk", None)
    return roof, roof_c


def bundlefusion_leg(args, dev, steps=10, warmup=3):
    """BASELINE.json configs[3]: BundleFusion indoor scene, 640x480, sphere 960x720 (train_bundlefusion.py:46-47), 96 samples/ray
    (U=64, G=4, P=8), R = 1080 rays per step (train_bundlefusion.py:32), D = 12 m -- the same training step as the headline."""
    from scenerf_amd.model import SceneRFBundleFusion
    R, U, P = 1080, 64, 8
    m = SceneRFBundleFusion(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=960, sphere_H=720, n_pts_uni=U, n_pts_per_gaussian=P,
                            max_sample_depth=12, precision=args.precision, device_rng=not args.host_rng).to(dev)
    m.mlp.load_state_dict(synth.mlp_state(11, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(12, 2, out_scale=0.5))
    opt = make_optimizer(args, list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters()))
    maps = {}
    for k, v in synth.feature_maps(960, 720, 13).items():
        if args.maps == "hwc":
            c, h, w = v.shape
            maps[k] = torch.empty_strided((c, h, w), (1, w * c, c), dtype=torch.float32, device=dev).copy_(v.to(dev)).requires_grad_(True)
        else:
            maps[k] = v.to(dev).requires_grad_(True)
    K, T = synth.bundlefusion_cam_K().to(dev), synth.rel_pose(0.3, 8.0).to(dev)
    pix = synth.stride2_pixels((640, 480), R, 14).to(dev)
    loss_fn = make_loss(args, dev, (640, 480), K, pix, 0, weights=(5.0, 1.0, 0.1))      # scenerf_bf.py:215,238

    def step():
        for v in maps.values():
            v.grad = None
        out = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=R)
        loss = loss_fn(out)
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(last).item()
    value = R / dt
    roof, roof_c = _leg_roofline(step, 2, args, U + 4 * P, value, R * (U + 4 * P))
    if roof:
        roof.pop("dense_equivalent", None)
    return {"metric": "rays/sec (training fwd+bwd), BundleFusion 640x480, 96 samples/ray", "value": round(value, 1), "unit": "rays/s",
            "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup,
            "config": {"workload": "BundleFusion 640x480, sphere 960x720, 96 samples/ray (U=64,G=4,P=8), 1080 rays/step, D=12 m, same step as the headline",
                       "rays_per_gpu": R, "samples_per_ray": U + 4 * P, "precision": args.precision, "maps": args.maps},
            "roofline": roof, "roofline_composite": roof_c}


def default_n64_leg(args, dev, steps=10, warmup=3):
    """The reference's own CLI default (train_kitti.py:33-35: n_pts_uni = 32, n_pts_per_gaussian = 8 -> 64 samples per ray), R = 1200 rays per
    step on the KITTI geometry: the same training step as the headline at half the samples (76,800 rows: 600 row blocks)."""
    import copy
    a2 = copy.copy(args)
    a2.samples = 64
    R = args.rays
    m = make_model(a2, dev)
    opt = make_optimizer(args, list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters()))
    maps = _make_maps(args.maps, dev, 0)
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
    pix = synth.stride2_pixels((1220, 370), R, 100).to(dev)
    loss_fn = make_loss(args, dev, (1220, 370), K, pix, 0)

    def step():
        for v in maps.values():
            v.grad = None
        loss = loss_fn(m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=R))
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(last).item()
    roof, roof_c = _leg_roofline(step, 2, args, 64, R / dt, R * 64)
    if roof:
        roof.pop("dense_equivalent", None)
    return {"metric": "rays/sec (training fwd+bwd), KITTI, 64 samples/ray (the reference's CLI default)", "value": round(R / dt, 1), "unit": "rays/s",
            "ms_per_step": round(dt * 1e3, 3), "steps": steps, "warmup": warmup, "issued": "eager",
            "config": {"workload": "KITTI 370x1220, sphere 1500x452, 64 samples/ray (U=32,G=4,P=8), %d rays/step, same step as the headline" % R,
                       "rays_per_gpu": R, "samples_per_ray": 64, "precision": args.precision, "maps": args.maps},
            "roofline": roof, "roofline_composite": roof_c}


class _StaticEncoder(torch.nn.Module):
    """Stands in for the image encoder of the trainer's step (net_rgb: EfficientNet-B7 U-Net, out of scope here and absent offline): hands
    back the same five feature maps with a batch axis -- they are leaves that require gradients, so the step computes the map gradients an
    encoder's backward would start from."""

    def __init__(self, maps):
        super().__init__()
        self.maps = maps

    def forward(self, img, pix=None, pix_sphere=None):
        return {k: v.unsqueeze(0) for k, v in self.maps.items()}


def kitti_training_step_leg(args, dev, steps=20, warmup=3, n_sources=2):
    """The step the reference's TRAINER runs (scenerf.py:119-241 through scenerf_amd.training.TrainingMixin.forward): one
    image, S source frames -- per source a trained render of n_rays = 1,200 rays, a metric-only render of 1,200 lidar pixels under no_grad and
    the source's loss -- then ONE optimizer step; issued eagerly (what Lightning's loop gets) and as ONE hipGraph replay
    (scenerf_amd.graph.GraphedFn, the per-source pixel subsets drawn on the device inside the graph).  rays/s counts the TRAINED rays."""
    from scenerf_amd.graph import GraphedFn
    R, S = args.rays, n_sources
    maps = {k: v.to(dev).requires_grad_(True) for k, v in synth.feature_maps(1500, 452, 3).items()}

    def build():
        m = make_model(args, dev)
        m.n_rays = R
        m.net_rgb = _StaticEncoder(maps)
        m.device_pixel_draw = True
        return m

    g = torch.Generator().manual_seed(7)
    img = lambda: torch.rand(3, 370, 1220, generator=g).to(dev)          # noqa: E731
    K = synth.kitti_cam_K().to(dev)
    batch = {"img_inputs": torch.rand(1, 3, 370, 1220, generator=g).to(dev), "cam_K": [K], "T_velo_2_cam": [torch.eye(4, device=dev)],
             "img_sources": [[img() for _ in range(S)]], "img_targets": [[img() for _ in range(S)]],
             "T_source2targets": [[synth.rel_pose(0.5 + 0.5 * i, 2.0).to(dev) for i in range(S)]],
             "T_source2infers": [[synth.rel_pose(1.0 + i, 0.0).to(dev) for i in range(S)]],
             "loc2d_with_depths": [[synth.stride2_pixels((1220, 370), R, 300 + i).to(dev) for i in range(S)]],
             "lidar_depths": [[torch.rand(R, generator=g).to(dev) * 60 + 2 for _ in range(S)]]}

    def timed(step):
        for _ in range(warmup):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            last = step()
        host = time.perf_counter() - t0
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, host / steps, last

    m = build()
    params = list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters())
    opt = make_optimizer(args, params)

    def eager_step():
        for v in maps.values():
            v.grad = None
        loss = m.step(batch, "train")
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss.detach()

    dt_e, host_e, last = timed(eager_step)
    assert torch.isfinite(last).item()
    out = {"metric": "rays/sec (the trainer's per-image step: %d source frames x (%d trained + %d metric-only rays), one optimizer step)" % (S, R, R),
           "unit": "rays/s", "sources": S, "trained_rays_per_step": S * R, "metric_only_rays_per_step": S * R, "steps": steps,
           "eager": {"value": round(S * R / dt_e, 1), "ms_per_step": round(dt_e * 1e3, 3), "host_issue_ms_per_step": round(host_e * 1e3, 3)}}
    del m, opt
    torch.cuda.empty_cache()
    graphed = None
    try:
        m2 = build()
        args.capturable = True
        opt2 = make_optimizer(args, list(m2.mlp.parameters()) + list(m2.mlp_gaussian.parameters()))
        args.capturable = False
        gs = GraphedFn(m2, opt2, lambda: m2.step(batch, "train"), dev, grad_leaves=list(maps.values()), warmup=warmup)
        dt_g, host_g, last = timed(gs)
        assert torch.isfinite(last).item()
        graphed = {"value": round(S * R / dt_g, 1), "ms_per_step": round(dt_g * 1e3, 3), "host_issue_ms_per_step": round(host_g * 1e3, 3)}
    except Exception as e:      # noqa: BLE001 -- the eager numbers stand on their own
        args.capturable = False
        graphed = {"error": repr(e)[:300]}
    out["graphed"] = graphed
    out["value"] = graphed.get("value", out["eager"]["value"]) if graphed else out["eager"]["value"]
    out["config"] = {"workload": "KITTI 370x1220, sphere 1500x452, %d samples/ray, 1 image x %d sources, %d trained + %d no_grad rays per source, "
                                 "scenerf_amd.training.TrainingMixin.forward + fused source loss + AdamW; encoder replaced by static maps" % (
                                     args.samples, S, R, R), "precision": args.precision}
    return out


def inference_leg(args, dev, frames=2, stride=2, chunk=4096, samples=512):
    """BASELINE.json configs[4]: KITTI novel-view inference, one stride-2 frame (112,850 px, generate_novel_depths.py:103-112), 512
    samples/ray (U=256, G=4, P=64), static chunks of 4,096 rays replayed from one captured hipGraph (scenerf_amd/inference.py)."""
    import copy
    from scenerf_amd.inference import pixel_grid
    a2 = copy.copy(args)
    a2.samples = samples
    model = make_model(a2, dev).eval()
    maps = {k: v.to(dev) for k, v in synth.feature_maps(1500, 452, 3).items()}
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
    grid = pixel_grid((1220, 370), stride, dev)
    n = grid.shape[0]

    def frame(graph=True):
        with torch.no_grad():
            return model.render_image(K, T, maps, sampled_pixels=grid, ray_batch_size=chunk, keys=("depth", "color"), use_graph=graph)

    frame()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(frames):
        last = frame()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    assert bool(torch.isfinite(last["depth"]).all())
    value = n / dt
    roof, roof_c = _leg_roofline(lambda: frame(False), 1, a2, samples, None, chunk * samples)
    model.release_inference_engine()
    U, P = sample_split(samples)
    return {"metric": "rays/sec (novel-view inference, %d samples/ray, hipGraph-replayed static chunks)" % samples, "value": round(value, 1),
            "unit": "rays/s", "ms_per_frame": round(dt * 1e3, 2), "frames": frames,
            "config": {"workload": "KITTI 370x1220 novel-view render, sphere 1500x452, %d px per frame (stride %d), %d samples/ray (U=%d,G=4,P=%d), "
                                   "chunks of %d rays (tail padded), no_grad, one hipGraph replayed per chunk" % (n, stride, samples, U, P, chunk),
                       "rays_per_frame": n, "samples_per_ray": samples, "chunk": chunk, "precision": args.precision,
                       "sampling_noise": "host generator per chunk" if args.host_rng else "device generator"},
            "roofline": roof, "roofline_composite": roof_c}


def graph_wanted(args, collectives):
    """(attempt a captured step?, reason if not).  One GPU and N > 1 alike: the timed step is ONE hipGraph replay per rank
    (scenerf_amd.graph.GraphedStep: forward + loss + backward with the gradient collectives + fused AdamW), attempted on every rank and
    kept only if every rank's capture succeeded (scenerf_amd.graph.build_on_all_ranks); otherwise all ranks issue the step eagerly."""
    if args.graph == "off":
        return False, "--graph off"
    if collectives and args.graph == "auto" and not getattr(args, "force_dist", False):
        # a replayed step whose all-reduces run BETWEEN GPUs has never executed on this code (one RCCL rank and two gloo ranks have): a
        # capture that raises is caught (build_on_all_ranks), a replay that hangs or reduces wrongly is not -- so more than one rank steps
        # eagerly unless `--graph on` asks for the captured step, which is then checked on its first replays (replay_agrees_across_ranks)
        return False, "more than one rank: the replayed step with its collectives is opt-in (--graph on) until it has run between GPUs"
    if args.sync == "step" and collectives:
        return False, "--sync step (its end-of-backward callback is host code)"
    if args.precision != "bf16" and args.graph == "auto":
        return False, "fp32 mode"
    if getattr(args, "host_rng", False):
        return False, "--host-rng (the host-side draw cannot be captured)"
    if args.optimizer != "fused":
        return False, "--optimizer torch"
    return True, None


def replay_agrees_across_ranks(tensors, world, dev=None):
    """First-replay check of a captured step that carries gradient collectives: after a replayed optimizer step every rank must
    hold the SAME parameters -- data-parallel ranks start equal and apply the same averaged gradients, so two checksums per rank (sum and
    sum of squares, float64) must be bit-equal everywhere.  A captured all-reduce that was not replayed, or reduced over the wrong ranks,
    leaves them different.  Every rank computes the same verdict from the gathered values: (ok, spread)."""
    acc = torch.zeros(2, dtype=torch.float64, device=tensors[0].device)
    for t in tensors:
        d = t.detach().double()
        acc[0] += d.sum()
        acc[1] += (d * d).sum()
    if world == 1:
        return True, 0.0
    got = [torch.zeros_like(acc) for _ in range(world)]
    torch.distributed.all_gather(got, acc)
    g = torch.stack(got)
    spread = float((g.max(0).values - g.min(0).values).abs().max())
    return spread == 0.0, spread


class _StubGraphed:
    """--dry-run stand-in for GraphedStep: 'captures' the stub step, or fails on the rank --dry-fail-capture names (tests/test_dist_gloo.py:
    the rank-consistent fallback of scenerf_amd.graph.build_on_all_ranks over gloo, without a GPU)."""

    def __init__(self, model, rank, fail_rank):
        if rank == fail_rank:
            raise RuntimeError("dry run: capture refused on rank %d" % rank)
        self.model = model

    def __call__(self):
        return self.model.step()


def main():
    args = parse()
    hook = os.environ.get("SRF_BENCH_TEST_ABORT")      # tests/test_dist_gloo.py: a child that dies like the HSA runtime's abort does
    if hook and os.environ.get("SRF_BENCH_CHILD"):
        if hook == "always" or not os.path.exists(hook):
            if hook != "always":
                open(hook, "w").close()
            os.abort()
    if args.dry_run:
        os.environ.setdefault("SRF_DIST_BACKEND", "gloo")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        # fail loud: a line that says n_gpus = WORLD_SIZE under a command that asked for --gpus N would be read as an N-GPU number
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to measure (launch with --nproc-per-node %d, or unset "
                         "WORLD_SIZE and let bench.py start its own ranks)" % (args.gpus, env_world, args.gpus))
    dry = args.dry_run
    if not dry:
        assert torch.cuda.is_available(), "bench.py needs a GPU"
        if torch.cuda.device_count() < env_world:
            raise SystemExit("bench.py: --gpus %d but only %d GPU(s) visible: one process per GPU is the contract" % (
                env_world, torch.cuda.device_count()))
    # (a captured step with collectives needs the NCCL watchdog's event polling off: scenerf_amd.dist.init_from_env)
    cap = graph_wanted(args, env_world > 1 or args.force_dist)[0] and (env_world > 1 or args.force_dist) and not dry
    rank, world, local = sdist.init_from_env(force=args.force_dist and not dry, graph_capture=cap)
    assert world == args.gpus
    forced = bool(args.force_dist and world == 1 and not dry)
    if forced:
        sdist.FORCE_COLLECTIVES = True
    pinned = None
    if not dry:
        dev = torch.device("cuda", local)
        torch.cuda.set_device(dev)
        if world > 1:
            pinned = sdist.numa_pin(local)      # launch calls, pinned buffers and the RCCL proxy thread on the GPU's own socket
        _capi.load()
        sync = torch.cuda.synchronize
    else:
        dev, sync = torch.device("cpu"), (lambda: None)
    census = _rank_census(rank, world, local, dev, dry)
    torch.manual_seed(42 + rank)  # train_kitti.py:11 seed_everything(42), decorrelated per rank
    if args.mode == "infer" and not dry:
        return infer_main(args, rank, world, dev, census)

    R = args.rays
    if dry:
        model, maps, K, T, pix, opt = _StubModel(rank), {}, None, None, None, None
    else:
        model = make_model(args, dev)
        # one source frame per step in every leg of this model: each step is a new image for the renderer.  The cache of converted maps
        # that serves the further source frames of ONE image (SceneRF.cache_converted_maps) would hit on the static bench maps every step:
        # off here, on (the default) in the trainer's multi-source leg
        model.cache_converted_maps = False
        params = list(model.mlp.parameters()) + list(model.mlp_gaussian.parameters())
        opt = make_optimizer(args, params)
        maps = _make_maps(args.maps, dev, rank)
        K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
        pix = synth.stride2_pixels((1220, 370), R, 100 + rank).to(dev)
    # N > 1: each MLP's packed gradient sink is all-reduced (RCCL) right before it is handed to autograd: 2 x 21.7 MB
    step_sync = None
    collectives = world > 1 or forced
    if collectives and args.sync == "step" and not dry:
        step_sync = sdist.StepGradSync(params)
    else:
        model.grad_sync = sdist.allreduce_mean_ if collectives else None
        model.grad_sync_async = sdist.allreduce_mean_async if collectives else None   # radiance MLP: started before the feature scatter
    loss_fn = make_loss(args, dev, (1220, 370), K, pix, rank) if not dry else None

    def make_step(model, opt):
        if dry:
            return model.step
        def step():
            for v in maps.values():
                v.grad = None
            out = model.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=R)
            loss = loss_fn(out)
            loss.backward()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss
        return step

    step = make_step(model, opt)
    # The timed step is ONE hipGraph replay per rank (the product's GraphedStep), on one GPU and on N alike: with its gradient collectives
    # captured (tests/test_gpu_graph.py::test_graphed_step_with_nccl_world1 runs that on RCCL with one rank; `--gpus 1 --force-dist` is the
    # same line through bench.py).  Every rank attempts its capture, then the ranks agree over a gloo side group: if ANY capture failed, ALL
    # ranks issue the step eagerly (scenerf_amd.graph.build_on_all_ranks; control flow under tests/test_dist_gloo.py with two gloo ranks) --
    # a rank-consistent fallback, never a mix and never a hang on a half-captured world.  One GPU: the eagerly issued step is measured
    # right after the replayed one and reported as `eager_step`.
    graphed = eager_leg = None
    graph_note = "eager (one launch call per kernel)"
    want_graph, why = graph_wanted(args, collectives)
    if want_graph:
        from scenerf_amd.graph import build_on_all_ranks
        if dry:
            factory = lambda: _StubGraphed(model, rank, args.dry_fail_capture)      # noqa: E731
        else:
            from scenerf_amd.graph import GraphedStep
            args.capturable = True
            opt_g = make_optimizer(args, params)
            args.capturable = False
            factory = lambda: GraphedStep(model, opt_g, loss_fn, K, T, maps, pix, ray_batch_size=R, warmup=max(1, args.warmup))     # noqa: E731
        graphed, note = build_on_all_ranks(factory)
        if graphed is not None and collectives and world > 1:
            # the replayed collectives are checked before anything is timed: two replays, then the parameters must be equal on every rank
            for _ in range(2):
                graphed()
            ok, spread = replay_agrees_across_ranks([model.head, model.main] if dry else params, world)
            if not ok:        # (every rank computed this from the same gathered values: all drop their graphs together)
                graphed, note = None, "replayed step left the ranks' parameters %.3e apart: all ranks step eagerly" % spread
            else:
                note += "; parameters bit-equal across ranks after the first replays"
        if graphed is not None:
            graph_note = "one hipGraph replay per step (scenerf_amd.graph.GraphedStep: forward + loss + backward%s + fused AdamW captured once; %s)" % (
                " + the gradient all-reduces" if collectives else "", note)
        else:
            graph_note = "eager (%s)" % note
            if not dry:
                torch.cuda.synchronize()
    else:
        graph_note = "eager (%s)" % why
    if not dry:
        for _ in range(max(0, args.device_warm_steps)):     # setup: the device at its sustained clocks before W + K (see --device-warm-steps)
            (graphed or step)()
    if graphed is not None:
        dt, last = _timed(graphed, args, world, dev, sync)
        host_ms = _timed.host_s / args.steps * 1e3
        if not args.headline_only:
            dt_e, last_e = _timed(step, args, world, dev, sync)
            assert torch.isfinite(last_e).item(), "loss is not finite (eager step)"
            eager_leg = {"value": round(world * R * args.steps / dt_e, 1), "unit": "rays/s", "ms_per_step": round(dt_e / args.steps * 1e3, 3),
                         "host_issue_ms_per_step": round(_timed.host_s / args.steps * 1e3, 3),
                         "note": "the same step issued eagerly (~50 launch calls per step from Python), same process, right after the timed region"}
        step_main = graphed
    else:
        dt, last = _timed(step, args, world, dev, sync)
        host_ms = _timed.host_s / args.steps * 1e3
        step_main = step
    ms = dt / args.steps * 1e3
    value = world * R * args.steps / dt
    assert torch.isfinite(last).item(), "loss is not finite"
    if dry:   # every rank holds the mean over ranks of both buffers
        want = 3.0 * sum(r + 1.0 for r in range(world)) / world
        assert abs(float(last) - want) < 1e-6, (float(last), want)

    steady = None
    _trace("timed region done: %.3f ms/step" % ms)
    if world == 1 and not dry and not args.no_roofline:
        # the same step over a longer window: 20 steps after 5 warm-ups start on a GPU that idled through the setup and measure 2-4 %
        # slower than a run of hundreds (same box); `value` stays the driver's K / W, this is the sustained rate next to it
        n_ss = 300
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_ss):
            step_main()
        torch.cuda.synchronize()
        d_ss = (time.perf_counter() - t0) / n_ss
        steady = {"value": round(R / d_ss, 1), "unit": "rays/s", "ms_per_step": round(d_ss * 1e3, 3), "steps": n_ss,
                  "note": "same step, same process, right after the timed region"}
    other = None
    _trace("other_entry")
    if world == 1 and not dry and not args.no_roofline:   # the same step with the other map layout at the boundary (side measurement)
        main_maps = maps
        maps = _make_maps("chw" if args.maps == "hwc" else "hwc", dev, rank)
        dt2, _ = _timed(step, args, world, dev, sync)
        other = {"maps": "chw" if args.maps == "hwc" else "hwc", "value": round(R * args.steps / dt2, 1), "unit": "rays/s",
                 "ms_per_step": round(dt2 / args.steps * 1e3, 3), "issued": "eager (compare with eager_step)"}
        maps = main_maps
        torch.cuda.empty_cache()
    rng_other = None
    if world == 1 and not dry and not args.no_roofline:   # ... and with the other source of the sampler's normal noise
        model.render_cfg.device_rng = not model.render_cfg.device_rng
        dt3, _ = _timed(step, args, world, dev, sync)
        rng_other = {"sampling_noise": "device generator" if model.render_cfg.device_rng else "host generator + upload, like the reference (utils.py:208-211)",
                     "value": round(R * args.steps / dt3, 1), "unit": "rays/s", "ms_per_step": round(dt3 / args.steps * 1e3, 3),
                     "issued": "eager (compare with eager_step)"}
        model.render_cfg.device_rng = not model.render_cfg.device_rng

    drop_in = None
    _trace("drop_in")
    if world == 1 and not dry and not args.no_roofline:
        # the configuration an UNMODIFIED reference caller presents: contiguous (C,H,W) maps (converted per call) AND the sampler's normal
        # noise drawn on the host generator and uploaded (utils.py:208-211) -- eager (that draw cannot be captured).  (A second
        # GraphedStep over the same parameters in this process is not attempted: their AccumulateGrad nodes stay bound to the first
        # capture's stream, the cross-stream hand-off autograd then inserts is illegal inside a capture, and the failure mode is a
        # segfault in hipStreamEndCapture, not an exception -- r04_c.)
        main_maps = maps
        maps = _make_maps("chw", dev, rank)
        rng0 = model.render_cfg.device_rng
        model.render_cfg.device_rng = False
        dt4, _ = _timed(step, args, world, dev, sync)
        model.render_cfg.device_rng = rng0
        drop_in = {"maps": "chw", "sampling_noise": "host generator + upload", "issued": "eager", "value": round(R * args.steps / dt4, 1),
                   "unit": "rays/s", "ms_per_step": round(dt4 / args.steps * 1e3, 3)}
        maps = main_maps
        torch.cuda.empty_cache()

    allreduce = None
    if collectives:   # three more steps with the collectives bracketed by events; every rank takes part, then the group is done
        sdist.TIMING = []
        for _ in range(3):
            step()
        allreduce = _allreduce_report(world, 3)
        sdist.TIMING = None
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() == world and (world > 1 or forced):
            # (5,541,892 floats: renderer.PackedMLP's sink of the radiance MLP; --dry-run: the stub's 4 KiB)
            allreduce["standalone"] = _collective_probe(world, dev, 1024 if dry else 5_541_892)
        sdist.verify_step_collectives()   # every rank issued the same number of gradient collectives (an error, not a hang, if not)

    roof = roof_c = None
    kernels = []
    _trace("roofline")
    # everything below is rank-0-only side measurement: no collective may be issued from here on (the other ranks are done)
    model.grad_sync = model.grad_sync_async = None
    if step_sync is not None:
        step_sync.close()
    if rank == 0 and not args.no_roofline:
        nprof = 3
        if dry:
            for _ in range(nprof):
                step()     # (the other ranks are gone: a collective here would hang -- the regression this mode exists for)
        else:
            lib = _capi.load()
            torch.cuda.synchronize()
            lib.scenerf_hip_profile_enable(1)
            for _ in range(nprof):
                step()
            torch.cuda.synchronize()
            kernels = _capi.profile_collect()
            lib.scenerf_hip_profile_enable(0)
            args.rows_per_launch = R * args.samples
            roof, roof_c = _rooflines(kernels, args, nprof, value / world, world)
        if args.kernels_json:
            with open(args.kernels_json, "w") as f:
                json.dump(kernels, f, indent=1)

    cpu = eager = fp32 = None
    _trace("fp32 / eager baselines")
    if dry:
        args.no_fp32_mode = args.no_eager_baseline = args.no_cpu_baseline = True
    if rank == 0 and world == 1 and args.precision == "bf16" and not args.no_fp32_mode:
        # the same step with fp32 MFMA end to end (RenderConfig.precision = "fp32": the per-layer GEMM path), so that the eager fp32
        # baseline below has a matched-precision partner
        try:
            m32 = make_model(args, dev, precision="fp32")
            o32 = make_optimizer(args, list(m32.mlp.parameters()) + list(m32.mlp_gaussian.parameters()))
            s32 = make_step(m32, o32)
            for _ in range(2):
                s32()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n32 = 5
            for _ in range(n32):
                s32()
            torch.cuda.synchronize()
            d32 = (time.perf_counter() - t0) / n32
            fp32 = {"value": round(R / d32, 1), "unit": "rays/s", "ms_per_step": round(d32 * 1e3, 3),
                    "kind": "this renderer with precision='fp32' (fp32 MFMA, per-layer GEMMs), same step, %d steps" % n32}
            del m32, o32, s32
        except Exception as e:  # never let the side measurement break the bench line
            fp32 = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_eager_baseline:
        try:
            eager = eager_gpu_baseline(args, dev)
            if fp32 and "value" in fp32:
                eager["speedup_at_matched_precision_fp32"] = round(fp32["value"] / eager["value"], 2)
            eager["speedup_bf16_vs_eager_fp32"] = round(value / eager["value"], 2)
        except Exception as e:  # never let the side measurement break the bench line
            eager = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    bf_leg = inf_leg = n64_leg = trainer_leg = None
    if rank == 0 and world == 1 and not dry and not args.no_extra_legs:
        legs = {}
        for name, fn in (("bf", bundlefusion_leg), ("infer", inference_leg), ("n64", default_n64_leg), ("trainer", kitti_training_step_leg)):
            _trace("leg %s" % name)
            try:
                legs[name] = fn(args, dev)
            except Exception as e:  # never let a side leg break the bench line
                legs[name] = {"error": repr(e)[:300]}
            torch.cuda.synchronize()
            torch.cuda.empty_cache()
        bf_leg, inf_leg, n64_leg, trainer_leg = legs["bf"], legs["infer"], legs["n64"], legs["trainer"]
    _trace("cpu baseline")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)
    _trace("line")

    if rank == 0:
        U, P = sample_split(args.samples)
        line = {
            "metric": "rays/sec (training fwd+bwd) at KITTI 128-sample config", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "host_issue_ms_per_step": round(host_ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "KITTI 370x1220, sphere 1500x452, %d samples/ray (U=%d,G=4,P=%d), %d rays/GPU/step, "
                                   "render_rays_batch fwd+bwd, %s, MLP packing, feature-map + MLP "
                                   "gradients, grad all-reduce (N>1), fused AdamW on both MLPs" % (
                                       args.samples, U, P, R,
                                       "feature maps handed over channels-last ((C,H,W) tensors with (H,W,C) memory, read in place; the "
                                       "contiguous-(C,H,W) entry with its per-call layout conversion is timed as other_entry)"
                                       if args.maps == "hwc" else "incl. map layout conversion (contiguous (C,H,W) maps)"),
                       "rays_per_gpu": R, "samples_per_ray": args.samples, "parallelism": "dp%d" % world, "grad_sync": (args.sync if collectives else None),
                       "optimizer": "scenerf_amd.optim.FusedAdamW" if args.optimizer == "fused" else "torch.optim.AdamW(fused=True)",
                       "precision": args.precision, "maps": args.maps, "step_issue": graph_note,
                       "sampling_noise": "host generator + upload, like the reference" if args.host_rng else
                                         "device generator (RenderConfig.device_rng=True; --host-rng gives the reference's host-side draw: +0.1-0.25 ms per step)",
                       "loss": "reference per-source loss (colour L1 + reprojection on synthetic images, KL, closest gaussian; scenerf_amd.loss_side."
                               "source_loss, one launch each way)" if args.loss == "source" else "proxy: four means in eager torch",
                       "numa_pin": pinned, "device_warm_steps": args.device_warm_steps,
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES", "runtime default (4)")},
            "eager_step": eager_leg, "other_entry": other, "other_rng": rng_other, "drop_in": drop_in, "steady_state": steady,
            "roofline": roof, "roofline_composite": roof_c, "cpu_baseline": cpu, "eager_gpu_baseline": eager, "fp32_mode": fp32,
            "bundlefusion_c4": bf_leg, "infer_c5": inf_leg, "kitti_default_n64": n64_leg, "kitti_training_step": trainer_leg,
            "allreduce": allreduce, "ranks": census,
        }
        # flat copies of the side measurements (scalars inside `config` / `roofline` survive the driver's `parsed` copy of this line; nested
        # objects and extra top-level keys do not).  drop_in_* = the configuration an unmodified reference caller presents.
        cfgd = line["config"]
        if drop_in:
            cfgd["drop_in_rays_per_s"], cfgd["drop_in_ms_per_step"] = drop_in["value"], drop_in["ms_per_step"]
            cfgd["drop_in_is"] = "contiguous (C,H,W) maps converted per call + host-generator noise uploaded per chunk, eager issue"
        if eager_leg:
            cfgd["eager_issue_rays_per_s"], cfgd["eager_host_issue_ms_per_step"] = eager_leg["value"], eager_leg["host_issue_ms_per_step"]
        if steady:
            cfgd["steady_state_rays_per_s"] = steady["value"]
        for nm, leg in (("bundlefusion_c4", bf_leg), ("infer_c5", inf_leg), ("kitti_default_n64", n64_leg), ("kitti_training_step", trainer_leg)):
            if leg and "value" in leg:
                cfgd[nm + "_rays_per_s"] = leg["value"]
        if trainer_leg and "eager" in trainer_leg:
            cfgd["kitti_training_step_eager_rays_per_s"] = trainer_leg["eager"]["value"]
            cfgd["kitti_training_step_eager_host_issue_ms"] = trainer_leg["eager"]["host_issue_ms_per_step"]
        if roof is not None:
            if roof_c:
                roof["frac_tail_hbm_at_this_chunk"] = roof_c["frac"]
                for k, v in roof_c.items():
                    if k.startswith("frac_65536_rays_"):
                        roof[k.replace("frac_65536_rays_", "frac_hbm_65536_rays_")] = v
            if inf_leg and inf_leg.get("roofline"):
                roof["frac_fwd_inference_c5"] = inf_leg["roofline"]["frac"]
            if eager and "value" in eager:
                roof["speedup_vs_eager_fp32_port_on_this_gpu"] = eager.get("speedup_bf16_vs_eager_fp32")
        if allreduce:
            cfgd["allreduce_max_exposed_wait_ms"] = allreduce["max_exposed_wait_ms_per_step"]
        if dry:
            line.update(metric="dry-run (control flow only, stub step over gloo)", dtype="none", data="none")
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
