"""scenerf_amd.graph.GraphedStep (one hipGraph per training step) and FusedAdamW(capturable=True): a replayed step is an eager step."""
import copy

import pytest
import torch

from scenerf_amd import synth
from scenerf_amd.model import SceneRF

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_capturable_adamw_follows_the_host_counted_one():
    from scenerf_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(3)
    shapes = [(512, 42), (512,), (4, 512), (4097,)]
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdamW(pa, lr=3e-3, betas=(0.9, 0.99), weight_decay=0.05)
    ob = FusedAdamW(pb, lr=3e-3, betas=(0.9, 0.99), weight_decay=0.05, capturable=True)
    sa, sb = torch.optim.lr_scheduler.ExponentialLR(oa, gamma=0.9), torch.optim.lr_scheduler.ExponentialLR(ob, gamma=0.9)
    for step in range(5):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).to(DEV)
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        sa.step(); sb.step()
    for a, b in zip(pa, pb):
        # (bias corrections in fp32 on the device against double on the host)
        torch.testing.assert_close(b.detach(), a.detach(), rtol=2e-5, atol=2e-6)
        assert float(ob.state[b]["step"]) == 5.0 and int(oa.state[a]["step"]) == 5
    # a parameter without a gradient cannot lag behind in this mode
    pb[1].grad = None
    with pytest.raises(RuntimeError, match="every parameter"):
        ob.step()
    # state_dict round trip: the device-side counter restarts from the loaded steps
    for b in pb:
        b.grad = torch.ones_like(b)
    pc = [torch.nn.Parameter(b.detach().clone()) for b in pb]
    oc = FusedAdamW(pc, lr=1.0, capturable=True)
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))
    for c in pc:
        c.grad = torch.ones_like(c)
    ob.step(); oc.step()
    for b, c in zip(pb, pc):
        assert torch.equal(b.detach(), c.detach())


def test_capturable_adamw_counts_its_steps_inside_the_update_kernel():
    """No increment launch: the kernel reads t (steps done), updates with t + 1, and its last workgroup to retire stores t + 1.  More
    than 48 tensors = two launches per step (only the last one counts), thousands of workgroups, replayed as a graph: after n steps the
    parameters equal the host-counted optimizer's and the device counter reads n, the scratch word 0."""
    from scenerf_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(512, 512)] * 20 + [(37,)] * 35
    pa = [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]
    pb = [torch.nn.Parameter(p.detach().clone()) for p in pa]
    oa = FusedAdamW(pa, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.01)
    ob = FusedAdamW(pb, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.01, capturable=True)
    grads = [torch.randn(s, generator=g).to(DEV) for s in shapes]
    for a, b, gr in zip(pa, pb, grads):
        a.grad, b.grad = gr.clone(), gr.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ob.step()
    torch.cuda.current_stream().wait_stream(side)
    oa.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        ob.step()
    for _ in range(6):
        graph.replay(); oa.step()
    torch.cuda.synchronize()
    assert float(ob._hyper[0][0][1]) == 7.0 and not ob._hyper[0][0][2:].any()
    for a, b in zip(pa, pb):
        torch.testing.assert_close(b.detach(), a.detach(), rtol=2e-5, atol=2e-6)


def test_load_state_dict_after_capture_keeps_the_captured_hyper_tensor():
    """A hipGraph that holds FusedAdamW(capturable=True).step() reads [lr, t] from a device tensor by ADDRESS: load_state_dict must refresh
    that tensor in place (loaded learning rate, loaded step count), not drop it -- later replays would count on freed memory."""
    from scenerf_amd.optim import FusedAdamW
    g = torch.Generator().manual_seed(4)
    p = torch.nn.Parameter(torch.randn(1000, generator=g).to(DEV))
    q = torch.nn.Parameter(p.detach().clone())
    opt, ref = FusedAdamW([p], lr=1e-2, weight_decay=0.0, capturable=True), FusedAdamW([q], lr=1e-2, weight_decay=0.0)
    grad = torch.randn(1000, generator=g).to(DEV)
    p.grad, q.grad = grad.clone(), grad.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        opt.step()
    torch.cuda.current_stream().wait_stream(side)
    ref.step()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        opt.step()
    graph.replay(); ref.step()
    torch.cuda.synchronize()
    ptr = opt._hyper[0][0].data_ptr()
    sd = copy.deepcopy(opt.state_dict())
    sd["param_groups"][0]["lr"] = 5e-3
    for st in sd["state"].values():
        st["step"] = torch.tensor(10.0, device=DEV)
    opt.load_state_dict(sd)
    assert opt._hyper[0][0].data_ptr() == ptr                      # same memory, new contents
    assert opt._hyper[0][0].tolist()[:3] == [pytest.approx(5e-3), 10.0, 0.0] and not opt._hyper[0][0][2:].any()    # [lr, steps done, the kernel's scratch]
    sr = copy.deepcopy(ref.state_dict())
    sr["param_groups"][0]["lr"] = 5e-3
    for st in sr["state"].values():
        st["step"] = 10
    ref.load_state_dict(sr)
    graph.replay(); ref.step()
    torch.cuda.synchronize()
    assert float(opt.state[p]["step"]) == 11.0 and int(ref.state[q]["step"]) == 11
    torch.testing.assert_close(p.detach(), q.detach(), rtol=2e-5, atol=2e-6)


def _setup(seed):
    torch.manual_seed(seed)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=32, n_pts_per_gaussian=8, precision="bf16",
                device_rng=True).to(DEV)
    m.mlp.load_state_dict(synth.mlp_state(1, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2, out_scale=4.0))
    from scenerf_amd.optim import FusedAdamW
    opt = FusedAdamW(list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters()), lr=1e-4, weight_decay=0.0, capturable=True)
    maps = {k: v.to(DEV).requires_grad_(True) for k, v in synth.feature_maps(1500, 452, 3).items()}
    K, T = synth.kitti_cam_K().to(DEV), synth.rel_pose(1.0, 0.0).to(DEV)
    pix = synth.stride2_pixels((1220, 370), 256, 100).to(DEV)
    g = torch.Generator().manual_seed(seed + 100)
    noise = (torch.rand(256, 32, generator=g).to(DEV), torch.randn(256, 4 * 8, generator=g).to(DEV))   # static: the same draw every step
    return m, opt, maps, K, T, pix, noise


def _loss(out):
    return out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()


def _eager_run(steps):
    m, opt, maps, K, T, pix, noise = _setup(5)
    losses = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        for v in maps.values():
            v.grad = None
        loss = _loss(m.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256, noise=noise))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return [p.detach().clone() for p in opt.param_groups[0]["params"]], {k: v.grad.detach().clone() for k, v in maps.items()}, losses


def _rel(a, b):
    num = sum(float((x.detach() - y.detach()).norm() ** 2) for x, y in zip(a, b)) ** 0.5
    return num / sum(float(x.detach().norm() ** 2) for x in a) ** 0.5


def test_graphed_step_replays_the_eager_step():
    """W warm-up steps + N replays of ONE captured step against W + N eager steps from the same start, with the sampler noise
    injected (the same static draw in every step of every run).  Two eager runs already differ -- bf16 weights, fp32 atomics in a
    different order, samples that land on the other side of a texel boundary -- so the yardstick is that run-to-run spread."""
    from scenerf_amd.graph import GraphedStep
    N, W = 3, 2
    ref, gref, losses = _eager_run(W + N)
    ref2, gref2, losses2 = _eager_run(W + N)
    m2, opt2, maps2, K2, T2, pix2, noise2 = _setup(5)
    gs = GraphedStep(m2, opt2, _loss, K2, T2, maps2, pix2, ray_batch_size=256, warmup=W, noise=noise2)
    got = [float(gs()) for _ in range(N)]
    torch.cuda.synchronize()
    assert all(torch.isfinite(torch.tensor(got)))
    assert float(opt2.state[opt2.param_groups[0]["params"][0]]["step"]) == float(W + N)     # every replay was an optimizer step
    m0 = _setup(5)[0]       # the starting point: the optimizer did move the parameters
    assert _rel(ref, list(m0.mlp.parameters()) + list(m0.mlp_gaussian.parameters())) > 1e-4
    spread_p = _rel(ref, ref2)
    # (+ 2e-4: the replayed kernels interleave differently from the eager ones, AdamW turns last-bit differences of tiny gradient entries
    #  into full-size steps of those entries -- see test_restore_true_first_replay_is_an_eager_first_step)
    assert _rel(ref, [p.detach() for p in opt2.param_groups[0]["params"]]) <= 3 * spread_p + 2e-4, (spread_p, losses, got)
    assert abs(got[-1] - losses[-1]) <= 3 * abs(losses2[-1] - losses[-1]) + 2e-3 * (1 + abs(losses[-1])), (losses, losses2, got)
    for k in gref:      # map gradients of the last replay land in the captured leaves' .grad
        spread = float((gref[k] - gref2[k]).norm())
        assert float((gref[k] - gs.map_grads[k]).norm()) <= 3 * spread + 2e-2 * float(gref[k].norm()) + 1e-6, (k, spread)


def test_graphed_step_with_the_device_draw_inside_the_graph():
    from scenerf_amd.graph import GraphedStep
    m, opt, maps, K, T, pix, _ = _setup(7)
    gs = GraphedStep(m, opt, _loss, K, T, maps, pix, ray_batch_size=256, warmup=2)
    got = [float(gs()) for _ in range(4)]
    torch.cuda.synchronize()
    assert all(torch.isfinite(torch.tensor(got)))
    assert len(set(round(x, 7) for x in got)) > 1, got      # a fresh draw per replay: the loss moves from step to step
    assert float(opt.state[opt.param_groups[0]["params"][0]]["step"]) == 6.0


def test_restore_true_first_replay_is_an_eager_first_step():
    """GraphedStep(restore=True): after construction (2 warm-up steps + the capture) parameters, AdamW moments and step count, the
    in-kernel sampler state and a listed extra tensor are what they were -- here an optimizer that had ALREADY stepped twice (a resumed
    run: its moments must come back, not zeros).  The first replay then equals the next eager step from the same state up to the run-to-run
    spread of two eager steps (fp32 atomics), the device draw included (same seed, same call counter)."""
    from scenerf_amd.graph import GraphedStep

    def start():
        m, opt, maps, K, T, pix, _ = _setup(9)
        m.reseed_device_rng(1234) if m.__dict__.get("_rng_states") else None
        st = m._device_rng_state(torch.device(DEV))
        st.copy_(torch.tensor([1234, 0, 0], dtype=torch.int64))
        for _ in range(2):                       # the optimizer has a history
            opt.zero_grad(set_to_none=True)
            for v in maps.values():
                v.grad = None
            _loss(m.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256)).backward()
            opt.step()
        torch.cuda.synchronize()
        return m, opt, maps, K, T, pix

    def eager_next(m, opt, maps, K, T, pix):
        opt.zero_grad(set_to_none=True)
        for v in maps.values():
            v.grad = None
        loss = _loss(m.render_rays_batch(K, T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256))
        loss.backward()
        opt.step()
        torch.cuda.synchronize()
        return float(loss), [p.detach().clone() for p in opt.param_groups[0]["params"]]

    la, pa = eager_next(*start())
    lb, pb = eager_next(*start())
    m, opt, maps, K, T, pix = start()
    params = opt.param_groups[0]["params"]
    before = dict(p=[p.detach().clone() for p in params], m=[opt.state[p]["exp_avg"].clone() for p in params],
                  v=[opt.state[p]["exp_avg_sq"].clone() for p in params], rng=m._device_rng_state(torch.device(DEV)).clone(),
                  hyper=opt._hyper[0][0].clone())
    extra = torch.tensor([7, 3], dtype=torch.int64, device=DEV)

    def loss_fn(out):
        extra.add_(1)                             # a tensor the step advances (stands for source_loss's [seed, calls])
        return _loss(out)

    gs = GraphedStep(m, opt, loss_fn, K, T, maps, pix, ray_batch_size=256, warmup=2, restore=True, restore_tensors=[extra])
    torch.cuda.synchronize()
    assert gs.steps_warmup == 0
    assert all(torch.equal(a, b.detach()) for a, b in zip(before["p"], params))
    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(before["m"], params)) and float(before["m"][0].abs().max()) > 0
    assert all(torch.equal(a, opt.state[p]["exp_avg_sq"]) for a, p in zip(before["v"], params))
    assert torch.equal(before["rng"][:2], m._device_rng_state(torch.device(DEV))[:2])
    assert torch.equal(before["hyper"], opt._hyper[0][0]) and float(opt.state[params[0]]["step"]) == 2.0
    assert extra.tolist() == [7, 3]
    lg = float(gs())
    torch.cuda.synchronize()
    assert float(opt.state[params[0]]["step"]) == 3.0 and extra.tolist() == [8, 4]
    # (two eager runs from one start agree to ~1e-8 -- same launch timing, same order of the fp32 atomics -- but the replayed step runs its
    #  kernels in another interleaving, and AdamW's m / sqrt(v) turns a last-bit difference of a tiny gradient entry into a full-size step
    #  of that entry: measured 2e-5 .. 4e-5 relative over all parameters, depending on the stream plumbing.  A replay that was NOT step 1
    #  from the caller's state -- stale moments, a step count off by one, the warm-up steps' parameters -- is off by >= 3e-3.)
    spread = _rel(pa, pb)
    assert _rel(pa, [p.detach() for p in params]) <= 3 * spread + 2e-4, (spread, _rel(pa, [p.detach() for p in params]))
    assert abs(lg - la) <= 3 * abs(la - lb) + 1e-3 * (1 + abs(la)), (la, lb, lg)


def test_graphed_step_draws_fresh_pixels_inside_the_graph():
    """``pixels`` as a callable: the reference's per-step draw (a random subset of the stride-2 grid, scenerf.py:253-264) made on the device
    INSIDE the captured step -- every replay renders another pixel set (torch's CUDA generator is graph-safe), with a static sampler noise the
    loss still moves from replay to replay, and the object exposes the last replay's pixels."""
    from scenerf_amd.graph import GraphedStep
    m, opt, maps, K, T, pix, noise = _setup(11)
    xs, ys = torch.arange(0, 1220, 2, device=DEV, dtype=torch.float32), torch.arange(0, 370, 2, device=DEV, dtype=torch.float32)
    grid = torch.stack(torch.meshgrid(xs, ys, indexing="ij"), dim=2).reshape(-1, 2)

    def draw():
        return grid[torch.randperm(grid.shape[0], device=DEV)[:256]]

    gs = GraphedStep(m, opt, _loss, K, T, maps, draw, ray_batch_size=256, warmup=2, noise=noise)
    seen, losses = [], []
    for _ in range(3):
        losses.append(float(gs()))
        torch.cuda.synchronize()
        seen.append(gs.pixels.clone())
    assert all(torch.isfinite(torch.tensor(losses)))
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2])
    assert len(set(round(x, 7) for x in losses)) > 1, losses
    assert all(bool(((p[:, 0] % 2 == 0) & (p[:, 1] % 2 == 0) & (p[:, 0] < 1220) & (p[:, 1] < 370)).all()) for p in seen)


def test_graphed_step_follows_new_intrinsics_copied_into_cam_K():
    """The inverse of cam_K is made on the HOST (SceneRF._inv_K) and lives in a device tensor the captured step holds by address; a replay
    runs no Python, so after ``cam_K.copy_(new)`` that tensor would keep the old inverse and every ray its old direction.
    ``GraphedStep.__call__`` refreshes it in place when cam_K's version counter has moved: the replay then renders what an eager call with
    the new intrinsics renders (optimizer-free steps: nothing else moves between the two)."""
    from scenerf_amd.graph import GraphedStep
    m, _, maps, K, T, pix, noise = _setup(13)
    K = K.clone()
    for v in maps.values():
        v.requires_grad_(False)
    outs = {}

    def keep(out):
        outs["depth"] = out["depth"]
        return _loss(out)

    gs = GraphedStep(m, None, keep, K, T, maps, pix, ray_batch_size=256, warmup=1, noise=noise)
    gs(); torch.cuda.synchronize()
    d_old = outs["depth"].detach().clone()
    inv_addr = m._inv_K_cache[2].data_ptr()
    K2 = K.clone(); K2[0, 0] *= 0.9; K2[1, 1] *= 0.9; K2[0, 2] += 7.0
    K.copy_(K2)
    gs(); torch.cuda.synchronize()
    d_new = outs["depth"].detach().clone()
    assert m._inv_K_cache[2].data_ptr() == inv_addr                                    # refreshed in place: the graph's address
    assert torch.equal(m._inv_K_cache[2].cpu(), torch.inverse(K2.cpu().float()))        # ... with the host's inverse of the new values
    with torch.no_grad():
        ref = m.render_rays_batch(K2.clone(), T, maps, T_cam2velo=None, sampled_pixels=pix, ray_batch_size=256, noise=noise)["depth"]
    assert not torch.allclose(d_old, d_new, rtol=1e-3), "the replay did not see the new intrinsics"
    torch.testing.assert_close(d_new, ref, rtol=2e-3, atol=2e-3)                        # (bf16 run-to-run level)


def test_restore_true_with_a_pixel_callable_gives_back_the_callers_generator_state():
    """restore=True snapshots torch's CUDA generator BEFORE the constructor's probe draw of a ``pixels`` callable (it was taken
    after it, so the first replay drew what an eager first step would have drawn second)."""
    from scenerf_amd.graph import GraphedStep
    m, opt, maps, K, T, pix, noise = _setup(15)
    xs, ys = torch.arange(0, 1220, 2, device=DEV, dtype=torch.float32), torch.arange(0, 370, 2, device=DEV, dtype=torch.float32)
    grid = torch.stack(torch.meshgrid(xs, ys, indexing="ij"), dim=2).reshape(-1, 2)

    def draw():
        return grid[torch.randperm(grid.shape[0], device=DEV)[:256]]

    torch.cuda.manual_seed(4242)
    state0 = torch.cuda.get_rng_state(torch.device(DEV))
    expect = draw().clone()                     # what an eager first step draws from the caller's generator state
    torch.cuda.set_rng_state(state0, torch.device(DEV))
    gs = GraphedStep(m, opt, _loss, K, T, maps, draw, ray_batch_size=256, warmup=2, noise=noise, restore=True)
    assert torch.equal(torch.cuda.get_rng_state(torch.device(DEV)), state0)
    gs(); torch.cuda.synchronize()
    assert torch.equal(gs.pixels, expect)


def _trainer_setup(seed, S=2, R=256):
    """The trainer's per-image step on a small shape: SceneRF with a static stand-in encoder, one image, S source frames."""
    torch.manual_seed(seed)
    torch.cuda.manual_seed(seed)
    m = SceneRF(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=32, n_pts_per_gaussian=8, precision="bf16",
                device_rng=True, n_rays=R).to(DEV)
    m.mlp.load_state_dict(synth.mlp_state(1, 4))
    m.mlp_gaussian.load_state_dict(synth.mlp_state(2, 2, out_scale=4.0))
    m._device_rng_state(torch.device(DEV)).copy_(torch.tensor([77, 0, 0], dtype=torch.int64))
    maps = {k: v.to(DEV).requires_grad_(True) for k, v in synth.feature_maps(1500, 452, 3).items()}

    class Enc(torch.nn.Module):
        def forward(self, img, pix=None, pix_sphere=None):
            return {k: v.unsqueeze(0) for k, v in maps.items()}

    m.net_rgb = Enc()
    m.device_pixel_draw = True
    from scenerf_amd.loss_side import make_rng_state
    object.__setattr__(m, "_loss_rng", make_rng_state(torch.device(DEV), seed=5))     # (the fused source loss's tie-breaking noise)
    g = torch.Generator().manual_seed(seed + 1)
    img = lambda: torch.rand(3, 370, 1220, generator=g).to(DEV)      # noqa: E731
    K = synth.kitti_cam_K().to(DEV)
    batch = {"img_inputs": torch.rand(1, 3, 370, 1220, generator=g).to(DEV), "cam_K": [K], "T_velo_2_cam": [torch.eye(4, device=DEV)],
             "img_sources": [[img() for _ in range(S)]], "img_targets": [[img() for _ in range(S)]],
             "T_source2targets": [[synth.rel_pose(0.5 + 0.5 * i, 2.0).to(DEV) for i in range(S)]],
             "T_source2infers": [[synth.rel_pose(1.0 + i, 0.0).to(DEV) for i in range(S)]],
             "loc2d_with_depths": [[synth.stride2_pixels((1220, 370), R, 300 + i).to(DEV) for i in range(S)]],
             "lidar_depths": [[torch.rand(R, generator=g).to(DEV) * 60 + 2 for _ in range(S)]]}
    from scenerf_amd.optim import FusedAdamW
    opt = FusedAdamW(list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters()), lr=1e-4, weight_decay=0.0, capturable=True)
    return m, opt, maps, batch


def test_graphed_fn_replays_the_trainers_multi_source_step():
    """scenerf_amd.graph.GraphedFn around the trainer's OWN per-image step (TrainingMixin.forward: two source frames, per source a trained
    render, a metric-only render under no_grad and the fused source loss; one optimizer step), the per-source pixel subsets drawn on the
    device inside the graph: W warm-up steps + N replays against W + N eager steps from the same seeds -- every generator involved (torch's
    CUDA generator for the pixels, the sampler's and the loss's in-kernel counters) advances per replay as it does per eager step, so the
    parameters must agree to the run-to-run level of the fp32 atomics (through AdamW: see test_restore_true_...)."""
    from scenerf_amd.graph import GraphedFn
    N, W = 3, 2

    def eager(steps):
        m, opt, maps, batch = _trainer_setup(31)
        losses = []
        for _ in range(steps):
            opt.zero_grad(set_to_none=True)
            for v in maps.values():
                v.grad = None
            loss = m.step(batch, "train")
            loss.backward()
            opt.step()
            losses.append(float(loss.detach()))
        torch.cuda.synchronize()
        return [p.detach().clone() for p in opt.param_groups[0]["params"]], losses

    pa, la = eager(W + N)
    pb, lb = eager(W + N)
    m, opt, maps, batch = _trainer_setup(31)
    gs = GraphedFn(m, opt, lambda: m.step(batch, "train"), DEV, grad_leaves=list(maps.values()), warmup=W)
    got = [float(gs()) for _ in range(N)]
    torch.cuda.synchronize()
    assert all(torch.isfinite(torch.tensor(got)))
    assert float(opt.state[opt.param_groups[0]["params"][0]]["step"]) == float(W + N)
    assert len(set(round(x, 6) for x in got)) > 1, got          # fresh pixels and noise per replay
    spread = _rel(pa, pb)
    rel = _rel(pa, [p.detach() for p in opt.param_groups[0]["params"]])
    assert rel <= 3 * spread + 2e-4, (spread, rel, la, got)
    # (the loss itself is a noisy observer: per-pixel min(reprojection, identity) over bf16 renders -- five runs of the same eager steps
    #  gave 1.0065 .. 1.0094 at the fifth step, replays 1.0041 .. 1.0085 (tools/trainer_cache_probe.py); a replay that drew OTHER pixels or
    #  noise than the eager step would be off by the step-to-step scale, 0.98 .. 1.11)
    assert abs(got[-1] - la[-1]) <= 3 * abs(lb[-1] - la[-1]) + 1.5e-2 * (1 + abs(la[-1])), (la, lb, got)
    assert max(abs(a - b) for a, b in zip(got, la[W:])) <= 3e-2, (la, got)
    assert all(v.grad is not None and bool(torch.isfinite(v.grad).all()) for v in maps.values())


def test_the_trainers_metric_only_sessions_reuse_the_packed_operands_of_the_step_and_nothing_older():
    """training.TrainingMixin.forward opens a scope in which the parameters cannot change and after which one backward runs: the S trained
    renders of the image are chunks of ONE session (shared gradient sinks), the S metric-only renders of another, which reads the operands
    the first packed (model.render_rays_batch, renderer.PackMLP).  Same first-step gradients (to the atomics' order), the same logged
    metrics bit for bit, the same loss as without the scope; the scope ends with ``forward`` (a no_grad render after an optimizer step
    packs again: it sees the new weights)."""
    import contextlib

    def run(scoped):
        m, opt, maps, batch = _trainer_setup(47)
        logged = {}
        m.log = lambda name, v, **k: logged.setdefault(name, []).append(v.detach().clone() if torch.is_tensor(v) else v)
        if not scoped:
            m._params_fixed = contextlib.nullcontext
        first = None
        for _ in range(2):
            opt.zero_grad(set_to_none=True)
            for v in maps.values():
                v.grad = None
            loss = m.step(batch, "train")
            loss.backward()
            if first is None:      # the first step's gradients: same parameters, same pixels and noise in both runs
                first = [p.grad.detach().clone() for p in opt.param_groups[0]["params"]] + [maps[k].grad.detach().clone() for k in sorted(maps)]
            opt.step()
        assert "_pack_cache" not in m.__dict__ and "_image_sessions" not in m.__dict__
        with torch.no_grad():
            after = m.render_rays_batch(batch["cam_K"][0], batch["T_source2infers"][0][0], {k: v.detach() for k, v in maps.items()},
                                        sampled_pixels=batch["loc2d_with_depths"][0][0], ray_batch_size=256)["depth"].clone()
        torch.cuda.synchronize()
        return float(loss.detach()), logged, after, first

    la, ga, aa, fa = run(True)
    lb, gb, ab, fb = run(False)
    # one session per image (S source frames accumulate into the same sinks) against one session per source frame + autograd's additions
    for x, y in zip(fa, fb):
        assert float((x - y).abs().max()) <= 2e-3 * float(y.abs().max()) + 1e-12, (x.shape, float((x - y).abs().max()), float(y.abs().max()))
    metric = lambda k: k.rsplit("/", 1)[-1] in ("abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3")      # noqa: E731
    assert set(ga) == set(gb) and any(metric(k) for k in ga)
    for k in ga:
        if metric(k):      # the metric-only renders (evaluate_depth): same operands, same kernels -> identical
            # (the first step's: the second step's parameters already carry the run-to-run order of the first backward's atomics)
            assert len(ga[k]) == 4 and all(torch.equal(x, y) for x, y in zip(ga[k][:2], gb[k][:2])), k
    assert abs(la - lb) <= 2e-3 * (1 + abs(lb))      # (the trained sessions are untouched; run-to-run atomics order through two AdamW steps)
    assert float((aa - ab).abs().max()) <= 2e-2 * float(ab.abs().max())


def test_validation_step_under_no_grad_takes_the_same_path_and_gives_the_same_loss():
    """Lightning's validation_step runs ``forward`` under torch.no_grad(): every render of the image is metric-only then (two no_grad
    sessions: the trained renders' pixels and the lidar pixels, the latter on the second stream), nothing is kept for a backward.  Same
    seeds, same parameters: the total equals the training forward's (the no_grad instantiation of the radiance forward rounds where the
    training one does), and the scope is gone afterwards."""
    def run(grad):
        m, opt, maps, batch = _trainer_setup(53)
        logged = {}
        m.log = lambda name, v, **k: logged.setdefault(name, []).append(float(v))
        with torch.enable_grad() if grad else torch.no_grad():
            loss = m.step(batch, "train" if grad else "val")
        torch.cuda.synchronize()
        assert "_image_sessions" not in m.__dict__ and "_pack_cache" not in m.__dict__
        assert loss.requires_grad == grad
        return float(loss), {k.split("/", 1)[-1]: v for k, v in logged.items()}

    lt, gt_ = run(True)
    lv, gv = run(False)
    assert abs(lt - lv) <= 2e-5 * (1 + abs(lt)), (lt, lv)
    assert set(gt_) == set(gv)
    for k in gt_:
        assert all(abs(a - b) <= 1e-4 * (1 + abs(a)) for a, b in zip(gt_[k], gv[k])), k


def test_bundlefusion_trainer_step_shares_one_session_per_image():
    """scenerf_bf.py:124-247 through BundleFusionTrainingMixin.forward on the GPU (3 source frames of n_rays // grid^2 rays, depth metrics
    at the sampled pixels under a mask): the sources are chunks of one session -- same total and the same gradients of every parameter
    and every map (to the atomics' order) as with one session per source and autograd's additions."""
    import contextlib
    from scenerf_amd.model import SceneRFBundleFusion
    S, R = 3, 1024

    def run(scoped):
        torch.manual_seed(5)
        torch.cuda.manual_seed(5)
        m = SceneRFBundleFusion(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=960, sphere_H=720, n_pts_uni=64,
                                n_pts_per_gaussian=8, max_sample_depth=12, precision="bf16", device_rng=True, n_rays=R, sample_grid_size=2).to(DEV)
        m.mlp.load_state_dict(synth.mlp_state(11, 4))
        m.mlp_gaussian.load_state_dict(synth.mlp_state(12, 2, out_scale=0.5))
        m._device_rng_state(torch.device(DEV)).copy_(torch.tensor([9, 0, 0], dtype=torch.int64))
        maps = {k: v.to(DEV).requires_grad_(True) for k, v in synth.feature_maps(960, 720, 13).items()}

        class Enc(torch.nn.Module):
            def forward(self, img, pix=None, pix_sphere=None):
                return {k: v.unsqueeze(0) for k, v in maps.items()}

        m.net_rgb = Enc()
        m.device_pixel_draw = True
        from scenerf_amd.loss_side import make_rng_state
        object.__setattr__(m, "_loss_rng", make_rng_state(torch.device(DEV), seed=5))
        if not scoped:
            m._params_fixed = contextlib.nullcontext
        g = torch.Generator().manual_seed(6)
        img = lambda: torch.rand(3, 480, 640, generator=g).to(DEV)      # noqa: E731
        K = synth.bundlefusion_cam_K().to(DEV)
        depth = lambda: (torch.rand(480, 640, generator=g) * 6 * (torch.rand(480, 640, generator=g) > 0.2)).to(DEV)      # noqa: E731
        batch = {"img_inputs": torch.rand(1, 3, 480, 640, generator=g).to(DEV), "cam_K_depth": [K],
                 "img_sources": [[img() for _ in range(S)]], "img_targets": [[img() for _ in range(S)]],
                 "T_source2targets": [[synth.rel_pose(0.1 + 0.1 * i, 3.0).to(DEV) for i in range(S)]],
                 "T_source2infers": [[synth.rel_pose(0.2 + 0.1 * i, 0.0).to(DEV) for i in range(S)]],
                 "source_depths": [[depth() for _ in range(S)]]}
        logged = {}
        m.log = lambda name, v, **k: logged.setdefault(name, []).append(float(v))
        loss = m.step(batch, "train")
        loss.backward()
        torch.cuda.synchronize()
        ps = list(m.mlp.parameters()) + list(m.mlp_gaussian.parameters())
        return float(loss), [p.grad.clone() for p in ps] + [maps[k].grad.clone() for k in sorted(maps)], logged

    la, ga, lga = run(True)
    lb, gb, lgb = run(False)
    assert abs(la - lb) <= 1e-5 * (1 + abs(lb)), (la, lb)                     # the forward is the same arithmetic
    assert set(lga) == set(lgb) and any(k.endswith("abs_rel") for k in lga)
    for x, y in zip(ga, gb):
        assert float((x - y).abs().max()) <= 2e-3 * float(y.abs().max()) + 1e-12, (x.shape, float((x - y).abs().max()), float(y.abs().max()))


def test_a_failed_capture_leaves_a_process_that_can_step_eagerly():
    """`build_on_all_ranks` promises: if a capture fails, the step is issued eagerly instead.  A capture that fails HALF-WAY (here: a loss that
    reads a value back to the host inside the captured step) must therefore leave no capture open, the caller's stream current, no stale
    HIP error and no side stream that was forked into it in use -- on ROCm every later launch of the process otherwise fails (round 6:
    tools/capture_abort_probe.py).  In its own process (tests/capture_abort_worker.py)."""
    import os, re, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "capture_abort_worker.py")], capture_output=True, text=True, timeout=600)
    m_ = re.search(r"CAPTURE_ABORT state_ok=(\w+) finite=(\w+)", r.stdout)
    assert r.returncode == 0 and m_, r.stdout[-1500:] + r.stderr[-1500:]
    assert m_.group(1) == "True" and m_.group(2) == "True", r.stdout[-500:]


def test_graphed_step_refuses_what_cannot_be_captured():
    from scenerf_amd.graph import GraphedStep
    from scenerf_amd.optim import FusedAdamW
    m, opt, maps, K, T, pix, _ = _setup(1)
    with pytest.raises(RuntimeError, match="capturable"):
        GraphedStep(m, FusedAdamW(list(m.mlp.parameters()), lr=1e-4), _loss, K, T, maps, pix)
    m.render_cfg.device_rng = False
    with pytest.raises(RuntimeError, match="device"):
        GraphedStep(m, opt, _loss, K, T, maps, pix)


def test_graphed_step_with_nccl_world1():
    """The gradient collectives on the real backend: a one-rank "nccl" (RCCL) process group on the test GPU, the renderer's hooks
    forced to issue their all-reduces (scenerf_amd.dist.FORCE_COLLECTIVES), eagerly and inside a captured GraphedStep
    (tests/nccl_worker.py).  A one-rank mean is the identity: every variant gives the hook-free step's gradients up to the run-to-run
    noise of fp32 atomics, and a replayed graph keeps stepping the optimizer."""
    import os, re, subprocess, sys
    here = os.path.dirname(os.path.abspath(__file__))
    import socket
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(here, "nccl_worker.py")], env=env, capture_output=True, text=True, timeout=600)
    err = re.search(r"NCCL_ERROR.*", r.stdout, re.S)
    what = re.findall(r"what\(\):.*", r.stderr)
    assert r.returncode == 0, (err.group(0)[-3000:] if err else (what[:2], r.stdout[-800:], r.stderr[-1500:]))
    mm = re.search(r"NCCL_RESULT spread=([\d.e+-]+) session=([\d.e+-]+) step=([\d.e+-]+) graph_steps=(\d+) moved=([\d.e+-]+) finite=(\w+)", r.stdout)
    assert mm, r.stdout[-2000:] + r.stderr[-2000:]
    spread, sess, step = float(mm.group(1)), float(mm.group(2)), float(mm.group(3))
    assert sess <= 3 * spread + 1e-6 and step <= 3 * spread + 1e-6, (spread, sess, step)
    assert int(mm.group(4)) == 5 and float(mm.group(5)) > 0 and mm.group(6) == "True"      # 2 warm-up steps + 3 replays
