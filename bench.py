#!/usr/bin/env python
"""bench.py -- rays/sec of the SceneRF training hot path (render_rays_batch fwd+bwd) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
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
oracle = a port of the reference, timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from scenerf_amd import _capi, synth  # noqa: E402
from scenerf_amd import dist as sdist  # noqa: E402
from scenerf_amd.model import SceneRF  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense MFMA bf16, MI355X_MICROARCH.md
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def pmc_traffic(logical_name):
    """HBM bytes per launch of the kernel FUNCTION that runs `logical_name`, from the committed rocprofv3 --pmc passes
    (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, tools/pmc_hbm.py).  PMC counters cannot be read from inside the
    process, so this is the last profiled value of the same bench command, not a live reading; None if absent."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
    except Exception:
        return None
    if logical_name.startswith("mlp_fwd_fused"):
        fn = "mlp_fused_kernel<0>"
    elif logical_name.startswith("mlp_bwd_fused"):
        fn = "mlp_fused_kernel<1>"
    elif "wgrad" in logical_name:
        fn = "gemm_tn_kernel"
    else:
        fn = "gemm_nt_glds_kernel" if logical_name.startswith("gemm_") and not logical_name.endswith("/g") else None
    if not fn:
        return None
    # the dominant launch of a logical kernel is the largest grid of its function (main MLP, not the 4-point gaussian head)
    sized = [(int(k.split("@grid=")[1]), k, v) for k, v in d.items() if k.startswith(fn) and "@grid=" in k]
    if sized and logical_name.startswith("mlp_"):
        _, k, v = max(sized)
        return {"bytes_per_launch": round(v["hbm_bytes_per_launch"]), "kernel_fn": k, "source": os.path.basename(files[-1]),
                "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, average over the launches of this grid size"}
    for k, v in d.items():
        if k.startswith(fn) and "@grid=" not in k:
            return {"bytes_per_launch": round(v["hbm_bytes_per_launch"]), "kernel_fn": k, "source": os.path.basename(files[-1]),
                    "note": "average over all launches of this kernel function in a step"}
    return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rays", type=int, default=1200, help="rays per GPU per step (reference n_rays)")
    ap.add_argument("--samples", type=int, default=128, choices=[64, 96, 128, 256, 512])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=1024, help="rays of the bounded CPU-baseline sample")
    ap.add_argument("--no-eager-baseline", action="store_true")
    ap.add_argument("--kernels-json", default="", help="write the per-kernel table here")
    return ap.parse_args()


def sample_split(n):
    # N = U + 4*P with U = N/2 (SURVEY §8d config mapping; 96 -> U=64,P=8)
    if n == 96:
        return 64, 8
    return n // 2, n // 8


def make_model(args, dev):
    U, P = sample_split(args.samples)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=U, n_pts_per_gaussian=P,
                precision=args.precision, device_rng=False).to(dev)
    m.mlp.load_state_dict(synth.mlp_state(1, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2, out_scale=4.0))
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
    """The oracle (port of the reference's eager path) on the host cores, bounded sample, fwd+bwd."""
    cores = os.cpu_count() or 1
    threads = min(cores, 32)   # torch's CPU kernels stop scaling (and regress) beyond a few dozen threads
    torch.set_num_threads(threads)
    run = _oracle_setup(args, "cpu")
    run(8, 50)  # warm-up (allocators, thread pool)
    R = args.cpu_rays
    t0 = time.perf_counter()
    run(R, 60)
    best = time.perf_counter() - t0
    return {"value": round(R / best, 2), "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": "%d rays x %d samples fwd+bwd on %d of %d host threads, full KITTI maps (%.1f s)" % (
                R, args.samples, threads, cores, best)}


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


def main():
    args = parse()
    rank, world, local = sdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if world != args.gpus and rank == 0:
        print("warning: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world), file=sys.stderr)
    dev = torch.device("cuda", local % torch.cuda.device_count())   # (modulo only matters for the 1-GPU gloo self-test)
    torch.cuda.set_device(dev)
    _capi.load()
    torch.manual_seed(42 + rank)  # train_kitti.py:11 seed_everything(42), decorrelated per rank

    R = args.rays
    model = make_model(args, dev)
    params = list(model.mlp.parameters()) + list(model.mlp_gaussian.parameters())
    opt = torch.optim.AdamW(params, lr=1e-5, weight_decay=0.0, fused=True)
    # N > 1: each MLP's packed gradient sink is all-reduced (RCCL) right before it is handed to autograd: 2 x 21.7 MB
    model.grad_sync = sdist.allreduce_mean_ if world > 1 else None
    model.grad_sync_async = sdist.allreduce_mean_async if world > 1 else None   # radiance MLP: started before the feature scatter
    maps = {k: v.to(dev).requires_grad_(True) for k, v in synth.feature_maps(1500, 452, 3 + rank).items()}
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
    pix = synth.stride2_pixels((1220, 370), R, 100 + rank).to(dev)

    def step():
        for v in maps.values():
            v.grad = None
        out = model.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=R)
        loss = out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
        loss.backward()
        opt.step()
        opt.zero_grad(set_to_none=True)
        return loss

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        last = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms = dt / args.steps * 1e3
    value = world * R * args.steps / dt
    assert torch.isfinite(last).item(), "loss is not finite"

    roof = roof_c = None
    kernels = []
    # everything below is rank-0-only side measurement: no collective may be issued from here on (the other ranks are done)
    model.grad_sync = model.grad_sync_async = None
    if rank == 0 and not args.no_roofline:
        lib = _capi.load()
        torch.cuda.synchronize()
        lib.scenerf_hip_profile_enable(1)
        nprof = 3
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        kernels = _capi.profile_collect()
        lib.scenerf_hip_profile_enable(0)
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
            roof = {"bound": "mfma", "kernel": dom["name"], "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(ach / peak, 4), "traffic": pmc_traffic(dom["name"]),
                    "avg_launch_us": round(dom["avg_us"], 2), "flops_per_launch": dom["flops"] / dom["launches"],
                    "all_mfma_kernels_achieved": round(tot_f / (tot_t * 1e-3) / 1e12, 2),
                    "all_mfma_kernels_frac": round(tot_f / (tot_t * 1e-3) / 1e12 / peak, 4),
                    "mfma_ms_per_step": round(tot_t / nprof, 3)}
            # with zero-K-block skipping, utilisation is reported on the FLOPs issued (above) and the rays/s are
            # quoted separately against the dense algorithmic count of the reference: fwd 2(N*5,405,696 + G*5,404,672), x3 fwd+bwd
            dense = 3.0 * 2.0 * (args.samples * 5405696 + 4 * 5404672)
            roof["dense_equivalent"] = {"flops_per_ray_fwd_bwd": dense, "achieved": round(value / world * dense / 1e12, 1),
                                        "unit": "TFLOP/s", "frac_of_peak": round(value / world * dense / 1e12 / peak, 4),
                                        "note": "rays/s x the reference's dense FLOPs per ray (it multiplies the out-of-range "
                                                "scales' zeros); not a utilisation: the kernels skip those K blocks"}
        comp = [k for k in kernels if k["name"] in ("composite_fwd", "composite_bwd")]
        if comp:
            b = sum(k["bytes"] for k in comp)
            t = sum(k["total_ms"] for k in comp)
            ach = b / (t * 1e-3) / 1e9
            roof_c = {"bound": "hbm", "kernel": "composite_fwd+composite_bwd", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                      "bytes_per_ray_fwd_bwd": 80 * args.samples + 64}
        if args.kernels_json:
            with open(args.kernels_json, "w") as f:
                json.dump(kernels, f, indent=1)

    cpu = eager = None
    if rank == 0 and world == 1 and not args.no_eager_baseline:
        try:
            eager = eager_gpu_baseline(args, dev)
        except Exception as e:  # never let the side measurement break the bench line
            eager = {"error": repr(e)[:200]}
        torch.cuda.empty_cache()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args)

    if rank == 0:
        U, P = sample_split(args.samples)
        line = {
            "metric": "rays/sec (training fwd+bwd) at KITTI 128-sample config", "value": round(value, 1), "unit": "rays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "KITTI 370x1220, sphere 1500x452, %d samples/ray (U=%d,G=4,P=%d), %d rays/GPU/step, "
                                   "render_rays_batch fwd+bwd incl. map layout conversion, MLP packing, feature-map + MLP "
                                   "gradients, grad all-reduce (N>1), fused AdamW on both MLPs" % (args.samples, U, P, R),
                       "rays_per_gpu": R, "samples_per_ray": args.samples, "parallelism": "dp%d" % world,
                       "precision": args.precision},
            "roofline": roof, "roofline_composite": roof_c, "cpu_baseline": cpu, "eager_gpu_baseline": eager,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
