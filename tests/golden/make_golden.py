#!/usr/bin/env python
"""Mint golden vectors from the REAL reference (run in the build container only).

The reference (astra-vision/SceneRF at /root/reference) has no tests or golden vectors
(SURVEY.md §4), so the pins for the oracle are produced here by importing its unmodified
modules and running ``SceneRF.render_rays_batch`` (+ autograd) on seeded synthetic inputs.
/root/reference does not exist on the GPU box, therefore only the *outputs* of this script
(tests/golden/*.npz) are used by the test-suite; this script is committed for provenance.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Shims (interface only, no arithmetic):
  * ``pytorch_lightning`` is absent offline -> 10-line stub package (LightningModule = nn.Module);
  * ``UNet2DSphere.build`` needs torch.hub/network -> returns an empty nn.Module (encoder is off-path);
  * the reference draws its sampling noise inside the hot path (torch.rand_like on device,
    utils.py:84; torch.normal on CPU, utils.py:208-211).  Both are monkey-patched for the
    duration of the call to return the seeded noise stored in the fixture, so oracle / HIP
    kernels can be fed the very same noise.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from scenerf_amd import synth  # noqa: E402

REF = "/root/reference"


def _install_reference():
    stub = tempfile.mkdtemp(prefix="pl_stub_")
    os.makedirs(os.path.join(stub, "pytorch_lightning"))
    with open(os.path.join(stub, "pytorch_lightning", "__init__.py"), "w") as f:
        f.write(
            "import torch.nn as nn\n"
            "class LightningModule(nn.Module):\n"
            "    def save_hyperparameters(self, *a, **k): pass\n"
            "    def log(self, *a, **k): pass\n"
            "    @property\n"
            "    def device(self): return next(self.parameters()).device\n"
            "class LightningDataModule: pass\n"
        )
    sys.path[:0] = [stub, REF]
    import scenerf.models.unet2d_sphere as U
    U.UNet2DSphere.build = classmethod(lambda cls, **kw: torch.nn.Module())


class InjectNoise:
    """Serve pre-generated noise to the reference's in-path RNG calls, chunk by chunk."""

    def __init__(self, noise_u, noise_g):
        self.noise_u, self.noise_g = noise_u, noise_g
        self.pu = self.pg = 0

    def __enter__(self):
        self._rand_like, self._normal = torch.rand_like, torch.normal

        def rand_like(t, *a, **k):
            n = t.shape[0]
            out = self.noise_u[self.pu:self.pu + n]
            assert out.shape == t.shape, (out.shape, t.shape)
            self.pu += n
            return out.clone()

        def normal(mean=None, std=None, *a, **k):
            n = mean.shape[0]
            out = self.noise_g[self.pg:self.pg + n]
            assert out.shape == mean.shape, (out.shape, mean.shape)
            self.pg += n
            return out.clone()

        torch.rand_like, torch.normal = rand_like, normal
        return self

    def __exit__(self, *exc):
        torch.rand_like, torch.normal = self._rand_like, self._normal


CASES = {
    # name: dict(variant, ctor kwargs, R, chunk, pose, seeds)
    "kitti_small_n64": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8,
                            sphere_W=376, sphere_H=114), R=24, chunk=24, pose=(2.0, 10.0), seed=101, smooth=False),
    "kitti_full_n64": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8),
                           R=16, chunk=16, pose=(1.0, 0.0), seed=202, smooth=False),
    "kitti_small_n128_chunks": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8,
                                    sphere_W=376, sphere_H=114, n_pts_uni=64, n_pts_per_gaussian=16),
                                    R=20, chunk=12, pose=(5.0, -10.0), seed=303, smooth=True),
    "bf_small_n96": dict(variant="bf", ctor=dict(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11,
                         sphere_W=240, sphere_H=180, n_pts_uni=64, n_pts_per_gaussian=8, max_sample_depth=12),
                         R=16, chunk=16, pose=(0.4, 10.0), seed=404, smooth=True),
    # above the fused-kernel threshold (>= 4096 rows per chunk): the bf16 path under test is the one bench.py times (fused forward,
    # fused dgrad chain), white-noise maps (a +-1 sphere index picks an unrelated texel: nothing hides index errors)
    "kitti_full_n128_r64": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8, n_pts_uni=64,
                                n_pts_per_gaussian=16), R=64, chunk=64, pose=(1.0, 0.0), seed=606, smooth=False),
    "bf_full_n96_r48": dict(variant="bf", ctor=dict(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=960,
                            sphere_H=720, n_pts_uni=64, n_pts_per_gaussian=8, max_sample_depth=12), R=48, chunk=48,
                            pose=(0.3, 8.0), seed=707, smooth=False),
    # the uniform-only branch (scenerf_bf.py:662-665: n_pts_uni == 0 and n_pts_per_gaussian == 1): the BundleFusion model draws 2 uniform
    # samples per ray (:623-626) and renders those alone; the gaussian head still feeds the KL term
    "bf_uniform_only": dict(variant="bf", ctor=dict(som_sigma=0.02, std=0.1, add_fov_hor=14, add_fov_ver=11, sphere_W=240, sphere_H=180,
                            n_pts_uni=0, n_pts_per_gaussian=1, max_sample_depth=12), R=32, chunk=20, pose=(0.4, 10.0), seed=808,
                            smooth=True),
    # BASELINE.json configs[0] ("4k rays x 64 samples, 4-layer 128-wide MLP, CPU forward only"): both MLPs replaced by
    # ResnetFC(d_in=42, n_blocks=1, d_hidden=128) = lin_in + (lin_z.0, fc_0, fc_1) + lin_out.  Forward only; the (R, N)
    # outputs are stored as digests to keep the fixture small.
    "c1_plumbing_r4096_n64": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8,
                                  sphere_W=376, sphere_H=114), R=4096, chunk=1024, pose=(2.0, -10.0), seed=505, smooth=True,
                                  mlp=dict(n_blocks=1, d_hidden=128), forward_only=True),
    # the same net shape TRAINED (round 6: scenerf_hip_resnetfc_backward): outputs and the gradient digests of every parameter and map from
    # the reference's autograd, and a second shape (2 blocks x 64) for the block loop
    "c1_train_r256_n64": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8,
                              sphere_W=376, sphere_H=114), R=256, chunk=256, pose=(2.0, -10.0), seed=515, smooth=True,
                              mlp=dict(n_blocks=1, d_hidden=128)),
    "generic_train_2x64_r96": dict(variant="kitti", ctor=dict(som_sigma=2.0, std=2.0, add_fov_hor=20, add_fov_ver=8,
                                   sphere_W=376, sphere_H=114), R=96, chunk=48, pose=(1.0, 5.0), seed=525, smooth=True,
                                   mlp=dict(n_blocks=2, d_hidden=64)),
}

OUT_KEYS = ["depth", "color", "gaussian_means", "gaussian_stds", "weights_at_depth", "closest_pts_to_depths",
            "loss_kl", "alphas", "som_vars", "densities", "weights", "depth_volumes"]


def grad_digest(g: torch.Tensor, k: int = 256):
    """Compact, order-independent digest of a big gradient: norm, sum, top-k (index, value)."""
    flat = g.reshape(-1)
    kk = min(k, flat.numel())
    top = torch.topk(flat.abs(), kk).indices
    top, _ = torch.sort(top)
    return dict(norm=float(flat.double().norm()), sum=float(flat.double().sum()),
                idx=top.numpy().astype(np.int64), val=flat[top].numpy().astype(np.float32))


def run_case(name, spec):
    if spec["variant"] == "kitti":
        from scenerf.models.scenerf import SceneRF as Model
        K = synth.kitti_cam_K()
    else:
        from scenerf.models.scenerf_bf import SceneRF as Model
        K = synth.bundlefusion_cam_K()
    model = Model(**spec["ctor"])
    seed = spec["seed"]
    mk = spec.get("mlp", {})
    if mk:
        from scenerf.models.resnetfc import ResnetFC
        model.mlp = ResnetFC(d_in=42, d_out=4, d_latent=2480, **mk)
        model.mlp_gaussian = ResnetFC(d_in=42, d_out=2, d_latent=2480, **mk)
    model.mlp.load_state_dict(synth.mlp_state(seed + 1, 4, **mk))
    model.mlp_gaussian.load_state_dict(synth.mlp_state(seed + 2, 2, out_scale=4.0, **mk))
    W, H = model.out_img_W, model.out_img_H
    x_rgb = synth.feature_maps(W, H, seed + 3, smooth=spec["smooth"])
    for v in x_rgb.values():
        v.requires_grad_(True)
    R = spec["R"]
    pix = synth.stride2_pixels(tuple(model.img_size), R, seed + 4)
    U, G, P = model.n_pts_uni, model.n_gaussians, model.n_pts_per_gaussian
    if U == 0 and spec["variant"] == "bf":
        U = 2    # scenerf_bf.py:623-626
    noise_u, noise_g = synth.sampling_noise(R, U, G * P, seed + 5)
    T = synth.rel_pose(*spec["pose"])
    with InjectNoise(noise_u, noise_g):
        kwargs = dict(ray_batch_size=spec["chunk"], sampled_pixels=pix)
        if spec["variant"] == "kitti":
            kwargs["T_cam2velo"] = torch.eye(4)
        out = model.render_rays_batch(K, T, x_rgb, **kwargs)
    loss = out["depth"].mean() + out["color"].mean() + out["loss_kl"].mean() + out["gaussian_means"].mean()
    fwd_only = spec.get("forward_only", False)
    if not fwd_only:
        loss.backward()

    blob = {"cam_K": K.numpy(), "T_source2infer": T.numpy(),
            "meta": np.array(repr(dict(variant=spec["variant"], ctor=spec["ctor"], R=R, chunk=spec["chunk"],
                                       seed=seed, smooth=spec["smooth"], sphere_W=W, sphere_H=H,
                                       img_size=tuple(model.img_size), mlp=mk, forward_only=fwd_only)))}
    if not fwd_only:   # (the big forward-only case regenerates pixels and noise from the seeds: golden_util.Golden)
        blob.update({"pixels": pix.numpy(), "noise_u": noise_u.numpy(), "noise_g": noise_g.numpy()})
    for k in OUT_KEYS:
        o = out[k].detach()
        if fwd_only and o.dim() == 2 and o.shape[1] > 8:     # (R, N) outputs of the big forward-only case: digest
            for kk, vv in grad_digest(o).items():
                blob["outdigest/%s/%s" % (k, kk)] = vv
        else:
            blob["out/" + k] = o.numpy()
    blob["loss"] = np.float64(loss.item())
    if fwd_only:
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **blob)
        print("%-28s loss=%.6f  depth[0:3]=%s  -> %s (%.1f KB, forward only)" % (
            name, loss.item(), out["depth"][:3].detach().numpy(), os.path.basename(path), os.path.getsize(path) / 1024))
        return
    for mod_name, mod in (("mlp", model.mlp), ("mlp_gaussian", model.mlp_gaussian)):
        for pn, p in mod.named_parameters():
            d = grad_digest(p.grad)
            for kk, vv in d.items():
                blob["grad/%s.%s/%s" % (mod_name, pn, kk)] = vv
    for key, v in x_rgb.items():
        g = v.grad if v.grad is not None else torch.zeros_like(v)
        d = grad_digest(g)
        for kk, vv in d.items():
            blob["grad/x_rgb.%s/%s" % (key, kk)] = vv
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **blob)
    print("%-28s loss=%.6f  depth[0:3]=%s  -> %s (%.1f KB)" % (
        name, loss.item(), out["depth"][:3].detach().numpy(), os.path.basename(path), os.path.getsize(path) / 1024))


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference is only mounted in the build container"
    torch.manual_seed(0)
    _install_reference()
    only = sys.argv[1:]
    for name, spec in CASES.items():
        if only and name not in only:
            continue
        run_case(name, spec)
