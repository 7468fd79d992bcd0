"""Worker of tests/test_gpu_graph.py::test_graphed_step_with_nccl_world1: ONE rank in an "nccl" (= RCCL) process group on the test box's
GPU.  With scenerf_amd.dist.FORCE_COLLECTIVES the renderer's gradient hooks issue their collectives although the group has a single
member, so the real backend's code path runs on one leased GPU: the communicator bound to the device (init_from_env's device_id), the
gaussian head's all-reduce in stream order on the backward's side stream, the radiance MLP's asynchronous all-reduce started before
the feature-gradient scatter and waited for in PackMLP.backward, StepGradSync's end-of-backward callback -- first issued eagerly, then
captured into scenerf_amd.graph.GraphedStep's hipGraph and replayed.  A one-rank mean is the identity, so every variant must give the
gradients of the step without hooks."""
import os, sys
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scenerf_amd import dist as sdist, synth   # noqa: E402
from scenerf_amd.model import SceneRF          # noqa: E402


def main():
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("LOCAL_RANK", "0")
    rank, world, local = sdist.init_from_env("nccl", force=True, graph_capture=True)
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
    sdist.FORCE_COLLECTIVES = True
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=32, n_pts_per_gaussian=8, precision="bf16",
                device_rng=True).to(dev)
    m.mlp.load_state_dict(synth.mlp_state(1, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2, out_scale=4.0))
    params = list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters())
    maps = {k: v.to(dev).requires_grad_(True) for k, v in synth.feature_maps(1500, 452, 3).items()}
    K, T = synth.kitti_cam_K().to(dev), synth.rel_pose(1.0, 0.0).to(dev)
    pix = synth.stride2_pixels((1220, 370), 256, 100).to(dev)
    g = torch.Generator().manual_seed(11)
    noise = (torch.rand(256, 32, generator=g).to(dev), torch.randn(256, 32, generator=g).to(dev))
    loss_fn = lambda out: out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()

    def grads(mode):
        m.grad_sync = sdist.allreduce_mean_ if mode == "session" else None
        m.grad_sync_async = sdist.allreduce_mean_async if mode == "session" else None
        ss = sdist.StepGradSync(params) if mode == "step" else None
        for p in params + list(maps.values()):
            p.grad = None
        n0 = sdist._ISSUED
        out = m.render_rays_batch(K, T, maps, sampled_pixels=pix, ray_batch_size=256, noise=noise)
        loss_fn(out).backward()
        torch.cuda.synchronize()
        issued = sdist._ISSUED - n0
        if ss is not None:
            assert ss.reductions == 1, ss.reductions
            ss.close()
        return torch.cat([p.grad.reshape(-1) for p in params]).clone(), issued

    g_plain, n_plain = grads("none")
    g_plain2, _ = grads("none")
    g_sess, n_sess = grads("session")
    g_step, _ = grads("step")
    spread = float((g_plain - g_plain2).norm() / g_plain.norm())        # (fp32 atomics: run-to-run noise of the same step)
    rel_sess = float((g_sess - g_plain).norm() / g_plain.norm())
    rel_step = float((g_step - g_plain).norm() / g_plain.norm())
    assert n_plain == 0 and n_sess == 2, (n_plain, n_sess)              # one all-reduce per MLP through the session hooks
    sdist.TIMING = []
    grads("session")
    timed = sorted(k for k, _, _ in sdist.TIMING)
    sdist.TIMING = None
    assert timed == ["sync", "wait"], timed                              # head in stream order, radiance MLP waited for late
    assert sdist.verify_step_collectives() >= 2

    # the whole step, collectives included, as ONE hipGraph
    from scenerf_amd.graph import GraphedStep
    from scenerf_amd.optim import FusedAdamW
    m.grad_sync, m.grad_sync_async = sdist.allreduce_mean_, sdist.allreduce_mean_async
    opt = FusedAdamW(params, lr=1e-4, weight_decay=0.0, capturable=True)
    before = torch.cat([p.detach().reshape(-1) for p in params]).clone()
    gs = GraphedStep(m, opt, loss_fn, K, T, maps, pix, ray_batch_size=256, warmup=2, noise=noise)
    losses = [float(gs()) for _ in range(3)]
    torch.cuda.synchronize()
    after = torch.cat([p.detach().reshape(-1) for p in params])
    moved = float((after - before).norm() / before.norm())
    steps = float(opt.state[params[0]]["step"])
    ok = all(torch.isfinite(torch.tensor(losses)))
    print("NCCL_RESULT spread=%.3e session=%.3e step=%.3e graph_steps=%.0f moved=%.3e finite=%s" % (spread, rel_sess, rel_step, steps, moved, ok), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except Exception:
        import traceback
        print("NCCL_ERROR " + traceback.format_exc(), flush=True)
        raise
