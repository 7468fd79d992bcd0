"""Caller side of the boundary: the reference's ``forward / training_step`` loss assembly around ``render_rays_batch``
(reference scenerf/models/scenerf.py:119-390, scenerf_bf.py for the indoor weights), in stock PyTorch.

This is NOT part of the accelerated hot path ("next"); it exists so a Lightning ``Trainer`` can call
``training_step`` on ``scenerf_amd.model.SceneRF`` exactly as it calls the reference.  Differences from the reference,
none of which change a loss value:
  * boolean-mask indexing (a device->host sync each) is replaced by masked means: mean over valid == sum(m*x)/sum(m);
  * the metric-only second render (scenerf.py:193-198) runs under ``no_grad`` (its graph is never back-propagated);
  * depth metrics are computed on the device with the same formulas as loss/depth_metrics.py (no ``.cpu().numpy()``).
"""
from __future__ import annotations

import contextlib
from typing import Dict

import torch
import torch.nn.functional as F


def sample_pix_features(pix: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """utils.py:250-266: bilinear sample of img (C,H,W) at pixel coords (B,2) -> (C,B)."""
    pix = pix.float()
    g = torch.stack([(pix[:, 0] / (img.shape[2] - 1) - 0.5) * 2, (pix[:, 1] / (img.shape[1] - 1) - 0.5) * 2], dim=1)
    out = F.grid_sample(img.unsqueeze(0), g.unsqueeze(0).unsqueeze(2).float(), align_corners=False, mode="bilinear",
                        padding_mode="zeros")
    return out.reshape(img.shape[0], -1)


def depth_errors(gt: torch.Tensor, pred: torch.Tensor, min_depth: float = 1e-3, max_depth: float = 80.0, mask=None):
    """loss/depth_metrics.py:3-24 on device: abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3.  With ``mask`` the means run over the
    masked entries only (== the reference's ``gt[mask], pred[mask]``, without the boolean indexing's host sync; all zeros if the
    mask is empty, where the reference skips the logging)."""
    if pred.is_cuda:
        # one launch (csrc/loss.hip: depth_errors_kernel) instead of ~35: the trainer evaluates this once per source frame
        from . import _capi
        gt_, pr_ = gt.detach().reshape(-1).float().contiguous(), pred.detach().reshape(-1).float().contiguous()
        mk = None if mask is None else mask.reshape(-1).to(torch.uint8).contiguous()
        if gt_.numel() != pr_.numel() or (mk is not None and mk.numel() != pr_.numel()):
            raise RuntimeError("depth_errors: gt, pred and mask must have the same number of entries")
        out = torch.empty(8, dtype=torch.float32, device=pred.device)
        with torch.cuda.device(pred.device):
            _capi.check(_capi.load().scenerf_hip_depth_errors(gt_.data_ptr(), pr_.data_ptr(), _capi.ptr(mk), pr_.numel(), float(min_depth),
                                                              float(max_depth), out.data_ptr(),
                                                              torch.cuda.current_stream(pred.device).cuda_stream), "depth_errors")
        return tuple(out[i] for i in range(7))
    pred = pred.clamp(min=min_depth, max=max_depth)
    if mask is None:
        mean = lambda t: t.float().mean()
    else:
        m = mask.to(pred.dtype)
        gt = torch.where(mask, gt, torch.ones_like(gt))          # keeps log / division finite where masked out
        den = m.sum().clamp(min=1.0)
        mean = lambda t: (t.to(pred.dtype) * m).sum() / den
    thresh = torch.maximum(gt / pred, pred / gt)
    a1, a2, a3 = [mean(thresh < 1.25 ** k) for k in (1, 2, 3)]
    rmse = torch.sqrt(mean((gt - pred) ** 2))
    rmse_log = torch.sqrt(mean((torch.log(gt) - torch.log(pred)) ** 2))
    abs_rel = mean(torch.abs(gt - pred) / gt)
    sq_rel = mean(((gt - pred) ** 2) / gt)
    return abs_rel, sq_rel, rmse, rmse_log, a1, a2, a3


class TrainingMixin:
    """forward / step / training_step / validation_step for SceneRF (KITTI) and SceneRFBundleFusion."""

    reproj_weight = 1.0        # scenerf_bf.py:215 uses 5.0
    dist2closest_weight = 0.01  # scenerf_bf.py:238 uses 0.1

    def _metric_max_depth(self) -> float:
        return 80.0            # compute_depth_errors' default (scenerf.py:325-328 passes none)

    def compute_reprojection_loss(self, pix_source, sampled_color_source, depth_rendered, img_target, inv_K, cam_K,
                                  T_source2target):
        """scenerf.py:349-386: min(L1 reprojection, L1 identity + 1e-5 noise) over pixels whose target point has z > 0."""
        homo = torch.cat([pix_source, torch.ones_like(pix_source[:, :1])], dim=1)
        cam_src = depth_rendered.reshape(-1, 1) * (inv_K @ homo.T).T
        h = torch.cat([cam_src, torch.ones_like(cam_src[:, :1])], dim=1)
        cam_tgt = (T_source2target @ h.T).T[:, :3]
        hp = (cam_K @ cam_tgt.T).T
        valid = cam_tgt[:, 2] > 0
        # masked division with a SAFE denominator: torch.where(z > 0, x / z, -1) alone still back-propagates 0/0 = NaN through the
        # unselected branch for a point with z == 0 (into depth_rendered and from there into every parameter); the reference
        # indexes with the mask before dividing (utils.py:308-313) and has no such path
        front = hp[:, 2] > 0
        z_safe = torch.where(front, hp[:, 2], torch.ones_like(hp[:, 2]))
        pix_tgt = torch.where(front[:, None], hp[:, :2] / z_safe[:, None], torch.full_like(hp[:, :2], -1.0))
        col_tgt = sample_pix_features(pix_tgt, img_target)
        col_id = sample_pix_features(pix_source, img_target)
        l_rep = torch.abs(col_tgt - sampled_color_source).mean(0)
        l_id = torch.abs(col_id - sampled_color_source).mean(0)
        l_id = l_id + torch.randn(l_id.shape, device=l_id.device) * 0.00001
        loss = torch.minimum(l_rep, l_id)
        m = valid.float()
        return (loss * m).sum() / m.sum().clamp(min=1.0)   # == loss[valid].mean()

    def _stride2_grid(self, dev, dtype):
        gkey = (dev, dtype, tuple(self.img_size))
        cache = self.__dict__.setdefault("_stride2_grids", {})
        grid = cache.get(gkey)
        if grid is None:      # the stride-2 pixel grid of scenerf.py:253-260: a function of the image size, built once per device
            xs = torch.arange(0, self.img_size[0], 2, device=dev, dtype=dtype)
            ys = torch.arange(0, self.img_size[1], 2, device=dev, dtype=dtype)
            gx, gy = torch.meshgrid(xs, ys, indexing="ij")
            grid = cache[gkey] = torch.stack([gx, gy], dim=2).reshape(-1, 2)
        return grid

    def _predraw(self, batch, n_rays, cam_K):
        """``device_pixel_draw``: the per-source pixel subsets of the whole batch, drawn before the first render -- the same calls on the
        same generator in the same order, only earlier: a randperm is a dozen small sort launches, and between two renders they wait for
        compute units behind the radiance forward of the metric-only render on the other stream (0.4 ms instead of 0.1 in the trace)."""
        if not getattr(self, "device_pixel_draw", False) or not cam_K.is_cuda:
            return
        n = self._stride2_grid(cam_K.device, cam_K.dtype).shape[0]
        self.__dict__["_predrawn_idx"] = [torch.randperm(n, device=cam_K.device)[:n_rays]
                                          for srcs in batch["img_sources"] for _ in srcs]

    def process_single_source(self, n_rays, x_rgb, cam_K, inv_K, img_source, img_target, T_source2target, T_source2infer,
                              T_cam2velo, step_type) -> Dict[str, torch.Tensor]:
        """scenerf.py:243-320."""
        dev = cam_K.device
        grid = self._stride2_grid(dev, cam_K.dtype)
        drawn = self.__dict__.get("_predrawn_idx")
        if drawn:
            idx = drawn.pop(0)       # (device_pixel_draw inside forward: drawn for all source frames of the batch up front, see _predraw)
        elif getattr(self, "device_pixel_draw", False):
            # the same draw on the DEVICE generator: no host round trip, and capturable (scenerf_amd.graph.GraphedFn replays the whole
            # per-image step with fresh pixels per replay: torch's CUDA generator is graph-safe)
            idx = torch.randperm(grid.shape[0], device=dev)[:n_rays]
        else:
            idx = torch.randperm(grid.shape[0])[:n_rays].to(dev)      # CPU generator like the reference (:262)
        pix_source = grid[idx]
        out = self.render_rays_batch(cam_K, T_source2infer, x_rgb, T_cam2velo=T_cam2velo,
                                     ray_batch_size=pix_source.shape[0], sampled_pixels=pix_source)
        depth, color = out["depth"], out["color"]
        self.log(step_type + "depth/closest_pts_to_depth", out["closest_pts_to_depths"].mean().detach(), on_epoch=True, sync_dist=True)
        self.log(step_type + "depth/weights_at_depth", out["weights_at_depth"].mean().detach(), on_epoch=True, sync_dist=True)
        if color.is_cuda and getattr(self, "fused_loss_side", True) and getattr(self, "fused_source_loss", True):
            # the WHOLE loss of this source frame in one kernel per direction (scenerf_amd.loss_side.source_loss -> csrc/loss.hip):
            # the three image gathers, the reprojection, both L1 terms, the closest-gaussian term, the means and the weights forward()
            # applies -- ~25 eager launches between the renderer's forward and its backward otherwise, each of them on the step's
            # critical path.  The tie-breaking noise (randn * 1e-5, scenerf.py:378) is made inside the kernel (Philox + Box-Muller on a
            # per-module seed and call counter); ``fused_loss_noise = "torch"`` draws it with the reference's torch.randn call instead
            from .loss_side import make_rng_state, source_loss
            noise = rng = None
            if getattr(self, "fused_loss_noise", "kernel") == "torch":
                noise = torch.randn(depth.shape[0], device=dev)     # the reference's call on the device generator (two more launches)
            else:   # made inside the kernel from a per-module (seed, call counter) pair in device memory
                rng = self.__dict__.get("_loss_rng")
                if rng is None or rng.device != dev:
                    rng = make_rng_state(dev)
                    object.__setattr__(self, "_loss_rng", rng)
            fused = source_loss(out, pix_source, img_source, img_target, cam_K, inv_K, T_source2target, noise=noise, noise_scale=0.00001,
                                reproj_weight=self.reproj_weight if self.use_reprojection else 0.0,
                                color_weight=1.0 if self.use_color else 0.0, dist2closest_weight=self.dist2closest_weight, rng_state=rng)
            terms = fused[1]
            self.log(step_type + "_som/dist_2_closest_gaussian", terms[4], on_epoch=True, sync_dist=True)
            self.log(step_type + "_som/closest_std", terms[6], on_epoch=True, sync_dist=True)
            return dict(fused=fused, depth_source_rendered=depth, pix_source=pix_source)
        diff = torch.abs(out["gaussian_means"] - depth.unsqueeze(-1).detach())
        min_diff, gi = torch.min(diff, dim=1)
        min_stds = torch.gather(out["gaussian_stds"], 1, gi.unsqueeze(-1))
        min_som_vars = torch.gather(out["som_vars"], 1, gi.unsqueeze(-1))
        self.log(step_type + "_som/dist_2_closest_gaussian", min_diff.mean().detach(), on_epoch=True, sync_dist=True)
        self.log(step_type + "_som/closest_std", min_stds.mean().detach(), on_epoch=True, sync_dist=True)
        if color.is_cuda and getattr(self, "fused_loss_side", True):
            # the three image gathers, the reprojection and both L1 terms in ONE kernel per direction (scenerf_amd/loss_side.py ->
            # csrc/loss.hip), consuming the renderer's depth / colour where they lie; the noise is drawn like the reference's
            from .loss_side import loss_side
            noise = torch.randn(depth.shape[0], device=dev) * 0.00001
            loss_color, loss_rep = loss_side(color, depth, pix_source, img_source, img_target, cam_K, inv_K, T_source2target, noise)
        else:   # CPU tensors (host-logic tests): the stock-PyTorch restatement, pinned on the reference's own functions
            col_src = sample_pix_features(pix_source, img_source)
            loss_color = torch.abs(color - col_src.T)
            loss_rep = self.compute_reprojection_loss(pix_source, col_src, depth, img_target, inv_K, cam_K, T_source2target)
        return dict(loss_kl=out["loss_kl"], loss_dist2closest_gauss=min_diff, loss_reprojection=loss_rep, loss_color=loss_color,
                    min_som_vars=min_som_vars, min_stds=min_stds, depth_source_rendered=depth, pix_source=pix_source)

    @staticmethod
    def _accumulate(tot, ret):
        """One source frame's terms into the running sums of forward() (scenerf.py:183-188).  With the fused source loss the frame's
        weighted total is already one differentiable scalar (``fused_total``) and the individual terms are only logged."""
        if ret.get("fused") is not None:
            total_src, terms = ret["fused"]
            tot["fused_total"] = total_src if "fused_total" not in tot else tot["fused_total"] + total_src
            # the logged terms as ONE vector (one launch per further source frame instead of six scalar additions per source)
            t = terms.detach()
            tot["fused_terms"] = t if "fused_terms" not in tot else tot["fused_terms"] + t
            return
        tot["somv"] = tot["somv"] + ret["min_som_vars"].mean()
        tot["kl"] = tot["kl"] + ret["loss_kl"].mean()
        tot["d2c"] = tot["d2c"] + ret["loss_dist2closest_gauss"].mean()
        tot["stds"] = tot["stds"] + ret["min_stds"].mean()
        tot["rep"] = tot["rep"] + ret["loss_reprojection"].mean()
        tot["col"] = tot["col"] + ret["loss_color"].mean()

    def _combine(self, tot, bs, step_type):
        """scenerf.py:203-238: the weighted total and the logged terms."""
        det = lambda t: t.detach() if torch.is_tensor(t) else t
        if "fused_terms" in tot:
            # every source frame's loss came out of the fused kernel: the weighted total is already one differentiable scalar, the logged
            # means are six entries of one vector (out8 of scenerf_hip_source_loss_forward: {total, rep, col, kl, d2c, som_vars, stds, valid})
            v = tot["fused_terms"] / bs
            if self.use_reprojection:
                self.log(step_type + "/loss_reprojection", v[1], on_epoch=True, sync_dist=True)
            if self.use_color:
                self.log(step_type + "/loss_color", v[2], on_epoch=True, sync_dist=True)
            self.log(step_type + "/loss_som_kl", v[3], on_epoch=True, sync_dist=True)
            self.log(step_type + "/min_som_vars", v[5], on_epoch=True, sync_dist=True)
            self.log(step_type + "/loss_dist2closest_gauss", v[4], on_epoch=True, sync_dist=True)
            total = tot["fused_total"] / bs
            self.log(step_type + "/total_loss", det(total), on_epoch=True, sync_dist=True)
            return {"total_loss": total}
        total = 0.0
        if self.use_reprojection:
            total = total + tot["rep"] / bs * self.reproj_weight
            self.log(step_type + "/loss_reprojection", det(tot["rep"] / bs), on_epoch=True, sync_dist=True)
        if self.use_color:
            total = total + tot["col"] / bs
            self.log(step_type + "/loss_color", det(tot["col"] / bs), on_epoch=True, sync_dist=True)
        total = total + tot["kl"] / bs
        self.log(step_type + "/loss_som_kl", det(tot["kl"] / bs), on_epoch=True, sync_dist=True)
        self.log(step_type + "/min_som_vars", det(tot["somv"] / bs), on_epoch=True, sync_dist=True)
        total = total + tot["d2c"] / bs * self.dist2closest_weight
        self.log(step_type + "/loss_dist2closest_gauss", det(tot["d2c"] / bs), on_epoch=True, sync_dist=True)
        if "fused_total" in tot:   # the same sum, assembled per source frame by the kernel (differentiable); the terms above were logged
            total = tot["fused_total"] / bs
        self.log(step_type + "/total_loss", det(total), on_epoch=True, sync_dist=True)
        return {"total_loss": total}

    # True: never synchronise for the depth metrics -- a source frame without a single valid depth then logs all-zero metrics (biasing
    # the on_epoch means towards 0); False (default): one scalar device->host read per masked evaluation decides whether to log at all,
    # like the reference's ``if mask.sum() > 0`` (scenerf_bf.py:204-205)
    sync_free_metrics = False

    def evaluate_depth(self, step_type, gt_depth, pred_depth, mask=None):
        """scenerf.py:322-346 (predictions clamped at the metric's default 80 m) / scenerf_bf.py:340-366 (at ``self.eval_depth``);
        metrics computed on device."""
        names = ["abs_rel", "sq_rel", "rmse", "rmse_log", "a1", "a2", "a3"]
        if mask is not None and not self.sync_free_metrics and not bool(mask.any()):
            return   # nothing to evaluate: the reference skips the logging (scenerf_bf.py:204-205)
        vals = depth_errors(gt_depth.reshape(-1).detach().float(), pred_depth.reshape(-1).detach().float(),
                            max_depth=self._metric_max_depth(), mask=None if mask is None else mask.reshape(-1))
        for n, v in zip(names, vals):
            self.log(step_type + "depth/" + n, v, on_epoch=True, sync_dist=True)

    @contextlib.contextmanager
    def _params_fixed(self):
        """The scope of one ``forward``: no optimizer step can happen inside and ONE backward follows (over the returned total), so the 2 S
        ``render_rays_batch`` calls of an image share what depends on the parameters and the image only: the S trained renders are chunks
        of one session and so are the S metric-only renders (model.render_rays_batch: one conversion, one pack, one set of gradient sinks),
        and the metric-only session reads the operands the trained one packed (renderer.PackMLP).  Everything dies with the scope."""
        self.__dict__["_pack_cache"] = {}
        self.__dict__["_image_sessions"] = {}       # model.render_rays_batch: the S source frames of an image render in ONE session
        try:
            yield
        finally:
            self.__dict__.pop("_pack_cache", None)
            self.__dict__.pop("_image_sessions", None)
            self.__dict__.pop("_predrawn_idx", None)

    # The metric-only renders (scenerf.py:190-201: a no_grad render of the lidar pixels per source frame, read by nothing but the depth
    # metrics) on a stream of their own, beside the trained renders: a render starts with ~0.3 ms of small dependent kernels (ray setup,
    # the gaussian head on 75 CUs, the sampler, encode + gather) before its MFMA-bound radiance forward -- issued beside another render's
    # radiance forward they cost nothing.  The sampler noise of that stream comes from a second call counter (model._device_rng_state).
    overlap_metric_renders = True

    # Below the default priority (HIP has such a level, 1; PyTorch's own streams stop at 0) so that the metric-only renders only take
    # compute units nobody waits for: measured on the trainer's step, same box, three alternations (profiles/r06_r_*) -- issued eagerly
    # 6.69-6.79 ms against 6.60-6.75 at the default priority, REPLAYED 8.75-9.32 ms against 6.41-6.60: a captured graph with a
    # low-priority branch loses the overlap altogether.  Off.
    metric_stream_low_priority = False

    def _metric_stream(self, dev):
        st = self.__dict__.setdefault("_metric_streams", {})
        key = torch.device(dev).index
        if key not in st:
            s = None
            if self.metric_stream_low_priority:
                # the metric-only renders should take the compute units nobody is waiting for: their radiance forward otherwise shares
                # the dispatcher round-robin with the trained render's small dependent kernels and doubles their latency
                import ctypes
                from . import _capi
                h, pr, low = ctypes.c_void_p(), ctypes.c_int(0), ctypes.c_int(0)
                with torch.cuda.device(dev):
                    _capi.check(_capi.load().scenerf_hip_stream_create_lowest_priority(ctypes.byref(h), ctypes.byref(pr), ctypes.byref(low)),
                                "stream_create_lowest_priority")
                if low.value:
                    s = torch.cuda.ExternalStream(h.value, device=dev)     # (lives as long as the process: never destroyed under pending work)
                else:
                    _capi.load().scenerf_hip_stream_destroy(h)
                self.__dict__["_metric_stream_priority"] = pr.value
            st[key] = s if s is not None else torch.cuda.Stream(device=dev)
        return st[key]

    def forward(self, batch, step_type):
        with self._params_fixed():
            return self._forward_batch(batch, step_type)

    def _forward_batch(self, batch, step_type):
        """scenerf.py:119-241.  ``self.net_rgb`` (stock encoder) must be set by the caller."""
        img_input = batch["img_inputs"]
        bs = img_input.shape[0]
        # scenerf.py:128 inverts batch["T_velo_2_cam"][0] here and threads the result down to predict() (:508), where nothing reads it;
        # torch.inverse checks LAPACK's info on the host (one device sync per step, and it cannot be captured), so the dead value is
        # not computed
        T_cam2velo = None
        cam_K0 = batch["cam_K"][0]
        pix, pix_sphere, _ = self.spherical_mapping.from_pixels(inv_K=self._inv_K(cam_K0))
        x_rgbs = self.net_rgb(img_input, pix=pix, pix_sphere=pix_sphere)
        tot = dict(rep=0.0, col=0.0, kl=0.0, somv=0.0, stds=0.0, d2c=0.0)
        if "_image_sessions" in self.__dict__:      # (inside forward's scope, which clears the list)
            self._predraw(batch, self.n_rays, cam_K0)
        side = None
        if self.overlap_metric_renders and "loc2d_with_depths" in batch and img_input.is_cuda:
            main = torch.cuda.current_stream(img_input.device)
            side = self._metric_stream(img_input.device)
            self._inv_K(batch["cam_K"][0])              # (sample 0's inverse intrinsics: uploaded in front of the event below)
            side.wait_event(main.record_event())        # the maps (and whatever made them) are in front of this point
        for i in range(bs):
            # (bs == 1, the trainers' batch size: the same view as x_rgbs[k][0], whose backward is a view too -- select's backward
            #  fills a zero (1,C,H,W) tensor and copies the gradient in: 0.84 GB of traffic per image at the KITTI shapes)
            x_rgb = {k: (x_rgbs[k].squeeze(0) if bs == 1 else x_rgbs[k][i]) for k in x_rgbs}
            cam_K = batch["cam_K"][i]
            inv_K = self._inv_K(cam_K)       # (the host's LAPACK: model.py)
            if side is not None and i > 0:
                # (the inverse of this sample's intrinsics may have been uploaded on the main stream just now: the second stream reads it
                #  too.  Sample 0's is in front of the first event; for further samples of a batch the second stream falls in behind the
                #  previous sample's renders here)
                side.wait_event(main.record_event())
            for sid in range(len(batch["img_sources"][i])):
                T_s2i = batch["T_source2infers"][i][sid]
                ret = self.process_single_source(self.n_rays, x_rgb=x_rgb, cam_K=cam_K, inv_K=inv_K,
                                                 img_source=batch["img_sources"][i][sid], img_target=batch["img_targets"][i][sid],
                                                 T_source2target=batch["T_source2targets"][i][sid], T_source2infer=T_s2i,
                                                 T_cam2velo=T_cam2velo, step_type=step_type)
                self._accumulate(tot, ret)
                if "loc2d_with_depths" in batch:   # depth metrics on the lidar pixels, scenerf.py:190-201
                    if side is None:
                        gt_pix = batch["loc2d_with_depths"][i][sid].float()
                        with torch.no_grad():
                            r = self.render_rays_batch(cam_K, T_s2i, x_rgb, ray_batch_size=gt_pix.shape[0], sampled_pixels=gt_pix)
                        self.evaluate_depth(step_type, batch["lidar_depths"][i][sid], r["depth"])
                    else:
                        self.__dict__["_rng_lane"] = 1
                        try:
                            with torch.cuda.stream(side), torch.no_grad():
                                # (EVERYTHING the second stream reads must have been made in front of an event it waited for, or be
                                #  made on it: the int -> float cast of the lidar pixels was a main-stream kernel once, and under load
                                #  the metric render read its output early -- found by running two test suites at once, round 6)
                                gt_pix = batch["loc2d_with_depths"][i][sid].float()
                                r = self.render_rays_batch(cam_K, T_s2i, x_rgb, ray_batch_size=gt_pix.shape[0], sampled_pixels=gt_pix)
                                self.evaluate_depth(step_type, batch["lidar_depths"][i][sid], r["depth"])
                        finally:
                            self.__dict__.pop("_rng_lane", None)
        if side is not None:
            main.wait_stream(side)       # the logged metrics are read behind this point
        return self._combine(tot, bs, step_type)

    def step(self, batch, step_type):
        return self.forward(batch, step_type)["total_loss"]

    def training_step(self, batch, batch_idx):
        return self.step(batch, "train")

    def validation_step(self, batch, batch_idx):
        self.step(batch, "val")


class BundleFusionTrainingMixin(TrainingMixin):
    """``forward`` of the indoor model for the BundleFusion collate layout (reference scenerf/models/scenerf_bf.py:124-247): one
    shared depth-camera intrinsic matrix, ``n_rays // sample_grid_size**2`` rays per source frame, depth metrics against the
    source frames' sensor depth at the sampled pixels, reprojection x5 and closest-gaussian x0.1 in the total.  Like the KITTI
    loop above it keeps the arithmetic and drops the host syncs (``if total > 0``, ``mask.sum() > 0``, boolean indexing)."""

    reproj_weight = 5.0          # scenerf_bf.py:215
    dist2closest_weight = 0.1    # scenerf_bf.py:238

    def _metric_max_depth(self) -> float:
        return float(self.eval_depth)   # scenerf_bf.py:347

    def _forward_batch(self, batch, step_type):
        if getattr(self, "smooth_loss_weight", 0) > 0:
            raise NotImplementedError("smooth_loss_weight > 0 calls compute_smooth_depth_loss, which the reference does not define")
        img_input = batch["img_inputs"]
        bs = img_input.shape[0]
        cam_K = batch["cam_K_depth"][0]
        inv_K = self._inv_K(cam_K)           # (the host's LAPACK: model.py)
        pix, pix_sphere, _ = self.spherical_mapping.from_pixels(inv_K=inv_K)
        x_rgbs = self.net_rgb(img_input, pix=pix, pix_sphere=pix_sphere)
        n_grids = self.n_rays // (self.sample_grid_size ** 2)
        tot = dict(rep=0.0, col=0.0, kl=0.0, somv=0.0, stds=0.0, d2c=0.0)
        if "_image_sessions" in self.__dict__:
            self._predraw(batch, n_grids, cam_K)
        for i in range(bs):
            x_rgb = {k: (x_rgbs[k].squeeze(0) if bs == 1 else x_rgbs[k][i]) for k in x_rgbs}
            for sid in range(len(batch["img_sources"][i])):
                ret = self.process_single_source(n_grids, x_rgb=x_rgb, cam_K=cam_K, inv_K=inv_K,
                                                 img_source=batch["img_sources"][i][sid], img_target=batch["img_targets"][i][sid],
                                                 T_source2target=batch["T_source2targets"][i][sid],
                                                 T_source2infer=batch["T_source2infers"][i][sid], T_cam2velo=None, step_type=step_type)
                self._accumulate(tot, ret)
                ps = ret["pix_source"].detach().long()
                depth_gt = torch.as_tensor(batch["source_depths"][i][sid]).to(ps.device)[ps[:, 1], ps[:, 0]]      # scenerf_bf.py:201-205
                self.evaluate_depth(step_type, depth_gt, ret["depth_source_rendered"], mask=depth_gt > 0)
        return self._combine(tot, bs, step_type)
