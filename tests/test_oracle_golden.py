"""Pin the CPU oracle against the golden vectors minted from the real reference."""
import pytest
import torch

import scenerf_oracle as orc
from golden_util import CASES, OUT_KEYS, Golden


def _cfg(g):
    base = orc.OracleConfig.kitti if g.variant == "kitti" else orc.OracleConfig.bundlefusion
    return base(**g.cfg_kwargs())


# (+ the two trainable generic-ResnetFC cases of round 6: BASELINE configs[0]'s 1 x 128 net and a 2 x 64 net, with the reference's autograd)
@pytest.mark.parametrize("name", CASES + ["c1_train_r256_n64", "generic_train_2x64_r96"])
def test_oracle_matches_reference_outputs_and_grads(name):
    g = Golden(name)
    cfg = _cfg(g)
    mlp, mlpg = g.mlp_states()
    for d in (mlp, mlpg):
        for v in d.values():
            v.requires_grad_(True)
    maps = g.feature_maps()
    for v in maps.values():
        v.requires_grad_(True)
    out = orc.render_rays_batch(cfg, mlp, mlpg, g.cam_K, g.T, maps, g.pixels, g.noise_u, g.noise_g,
                                ray_batch_size=g.chunk)
    for k in OUT_KEYS:
        ref = g.out(k)
        assert out[k].shape == ref.shape, k
        torch.testing.assert_close(out[k].detach(), ref, rtol=2e-5, atol=2e-6, msg=lambda m: "%s: %s" % (k, m))
    loss = orc.training_proxy_loss(out)
    assert abs(loss.item() - float(g.z["loss"])) < 1e-4 * abs(float(g.z["loss"]))
    loss.backward()
    tensors = {}
    for pn, p in mlp.items():
        tensors["mlp." + pn] = p.grad
    for pn, p in mlpg.items():
        tensors["mlp_gaussian." + pn] = p.grad
    for key, v in maps.items():
        tensors["x_rgb." + key] = v.grad if v.grad is not None else torch.zeros_like(v)
    assert set(g.grad_names()) == set(tensors)
    for nm, grad in tensors.items():
        d = g.grad_digest(nm)
        flat = grad.reshape(-1)
        assert abs(float(flat.double().norm()) - d["norm"]) <= 1e-4 * d["norm"] + 1e-9, nm
        torch.testing.assert_close(flat[d["idx"]], d["val"], rtol=1e-4, atol=1e-7,
                                   msg=lambda m: "%s: %s" % (nm, m))


def test_sort_is_a_permutation_and_sorted():
    g = Golden("kitti_small_n64")
    cfg = _cfg(g)
    mlp, mlpg = g.mlp_states()
    out = orc.render_chunk(cfg, mlp, mlpg, g.cam_K, g.T, g.feature_maps(), g.pixels, g.noise_u, g.noise_g,
                           keep_intermediates=True)
    perm = out["_perm"]
    assert torch.equal(torch.sort(perm, dim=1).values, torch.arange(perm.shape[1]).expand_as(perm))
    d = out["_dist_sorted"]
    assert bool((d[:, 1:] >= d[:, :-1]).all())


def test_oracle_matches_reference_on_the_plumbing_config():
    """BASELINE.json configs[0]: 4k rays x 64 samples through a 4-layer 128-wide ResnetFC (n_blocks=1), CPU forward only.  The golden
    vector was produced by the reference's own render_rays_batch with its mlp / mlp_gaussian swapped for that shape."""
    g = Golden("c1_plumbing_r4096_n64")
    assert g.meta["mlp"] == dict(n_blocks=1, d_hidden=128) and g.meta["R"] == 4096
    cfg = _cfg(g)
    mlp, mlpg = g.mlp_states()
    assert mlp["lin_in.weight"].shape == (128, 42) and "blocks.1.fc_0.weight" not in mlp
    with torch.no_grad():
        out = orc.render_rays_batch(cfg, mlp, mlpg, g.cam_K, g.T, g.feature_maps(), g.pixels, g.noise_u, g.noise_g,
                                    ray_batch_size=g.chunk)
    for k in OUT_KEYS:
        if ("out/" + k) in g.z.files:
            ref = g.out(k)
            assert out[k].shape == ref.shape, k
            torch.testing.assert_close(out[k], ref, rtol=2e-5, atol=2e-6, msg=lambda m: "%s: %s" % (k, m))
        else:
            d = g.out_digest(k)
            flat = out[k].reshape(-1)
            assert abs(float(flat.double().norm()) - d["norm"]) <= 2e-5 * d["norm"] + 1e-9, k
            torch.testing.assert_close(flat[d["idx"]], d["val"], rtol=2e-5, atol=2e-6, msg=lambda m: "%s: %s" % (k, m))
    assert abs(orc.training_proxy_loss(out).item() - float(g.z["loss"])) < 1e-4 * abs(float(g.z["loss"]))
