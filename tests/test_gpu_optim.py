"""scenerf_amd.optim.FusedAdamW (one HIP launch per parameter group) against torch.optim.AdamW: same trajectory, same state layout
(reference: configure_optimizers, scenerf.py:756-761 -- AdamW + ExponentialLR)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(512, 42), (512,), (4, 512), (4,), (512, 512), (512, 2480), (3, 5, 7), (1,), (4097,)]
    return [torch.nn.Parameter(torch.randn(s, generator=g).to(DEV)) for s in shapes]


@pytest.mark.parametrize("wd", [0.0, 0.05])
def test_fused_adamw_follows_torch_adamw(wd):
    from scenerf_amd.optim import FusedAdamW
    pa, pb = _params(1), _params(1)
    oa = torch.optim.AdamW(pa, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    ob = FusedAdamW(pb, lr=3e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=wd)
    sa = torch.optim.lr_scheduler.ExponentialLR(oa, gamma=0.95)
    sb = torch.optim.lr_scheduler.ExponentialLR(ob, gamma=0.95)
    gen = torch.Generator().manual_seed(2)
    for step in range(6):
        for i, (a, b) in enumerate(zip(pa, pb)):
            if step == 2 and i == 3:
                a.grad = b.grad = None          # a parameter without a gradient this step: skipped, its step count lags
                continue
            g = torch.randn(a.shape, generator=gen).to(DEV) * (0.1 + step)
            a.grad = g.clone()
            if i == 0:   # the renderer's lin_in.weight gradient: a column slice of a 256-wide sink, handed over as a view
                wide = torch.zeros(512, 256, device=DEV)
                wide[:, :42] = g
                b.grad = wide[:, :42]
                assert not b.grad.is_contiguous()
            else:
                b.grad = g.clone()
        oa.step(); ob.step()
        sa.step(); sb.step()
    for i, (a, b) in enumerate(zip(pa, pb)):
        torch.testing.assert_close(b.detach(), a.detach(), rtol=2e-6, atol=1e-6, msg=lambda m, i=i: "param %d: %s" % (i, m))
        sta, stb = oa.state[a], ob.state[b]
        assert int(stb["step"]) == int(sta["step"])
        # (moments of gradients of size ~0.5: last-ulp differences of the lerp / fma forms, absolute)
        torch.testing.assert_close(stb["exp_avg"], sta["exp_avg"], rtol=2e-6, atol=1e-6)
        torch.testing.assert_close(stb["exp_avg_sq"], sta["exp_avg_sq"], rtol=2e-6, atol=1e-6)
    # state_dict round trip into a fresh optimizer continues the same trajectory
    pc = [torch.nn.Parameter(b.detach().clone()) for b in pb]
    oc = FusedAdamW(pc, lr=1.0)
    import copy
    oc.load_state_dict(copy.deepcopy(ob.state_dict()))     # (load_state_dict itself may alias same-dtype, same-device state tensors)
    g = [torch.randn(b.shape, generator=gen).to(DEV) for b in pb]
    for b, c, gg in zip(pb, pc, g):
        b.grad, c.grad = gg.clone(), gg.clone()
    ob.step(); oc.step()
    for b, c in zip(pb, pc):
        assert torch.equal(b.detach(), c.detach())


def test_fused_adamw_refuses_what_it_does_not_implement():
    from scenerf_amd.optim import FusedAdamW
    p = torch.nn.Parameter(torch.zeros(8))          # CPU parameter: no fallback
    o = FusedAdamW([p], lr=1e-3)
    p.grad = torch.ones(8)
    with pytest.raises(RuntimeError, match="CUDA"):
        o.step()
    with pytest.raises(ValueError):
        FusedAdamW([torch.nn.Parameter(torch.zeros(2, device=DEV))], lr=-1.0)
