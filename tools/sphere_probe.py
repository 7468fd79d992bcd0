"""Image->sphere resampling at the KITTI sizes: per level, the HIP forward / backward gather kernels (in-library HIP
events) against their algorithmic bytes, next to the reference's own formulation run with eager torch on the same GPU (fresh
scatter + normalise + F.grid_sample + permute per call, unet2d_sphere.py:138-165).  usage: sphere_probe.py [reps]"""
import json, os, sys
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scenerf_amd import _capi
from scenerf_amd.model import SphericalMapping
from scenerf_amd.sphere import SphereResampler
from scenerf_amd.synth import kitti_cam_K

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
img_W, img_H, out_W, out_H = 1220, 370, 1500, 452
sm = SphericalMapping(img_W=img_W, img_H=img_H, out_img_W=out_W, out_img_H=out_H, v_angle_max=104.7294 + 8, v_angle_min=75.4815 - 8,
                      h_angle_max=131.1128 + 20, h_angle_min=49.5950 - 20)
K = kitti_cam_K().to(dev)
pix, pix_sphere, _ = sm.from_pixels(inv_K=torch.inverse(K))


def plane(scale):
    w, h, s = img_W, img_H, 1
    while s < scale:
        w, h, s = (w + 1) // 2, (h + 1) // 2, s * 2
    return w, h


def reference_form(x, scale):
    """unet2d_sphere.py:138-165, eager."""
    ow, oh = round(out_W / scale), round(out_H / scale)
    m = torch.zeros((ow, oh, 2)).type_as(x) - 10.0
    pss = torch.round(pix_sphere / scale).long()
    pss[:, 0] = pss[:, 0].clamp(0, ow - 1); pss[:, 1] = pss[:, 1].clamp(0, oh - 1)
    m[pss[:, 0], pss[:, 1], :] = pix // scale
    m = m.reshape(-1, 2)
    m[:, 0] /= x.shape[3]; m[:, 1] /= x.shape[2]
    m = (m * 2 - 1).reshape(1, 1, -1, 2)
    f = F.grid_sample(x, m, align_corners=False, mode="bilinear")
    return f.reshape(f.shape[0], f.shape[1], ow, oh).permute(0, 1, 3, 2)


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3      # us


rs = SphereResampler(out_W, out_H)
lib = _capi.load()
rows, tot = [], dict(hip_fwd=0.0, hip_bwd=0.0, ref_fwd=0.0, ref_fb=0.0, hip_fb=0.0)
for scale, C in ((1, 3), (2, 32), (4, 48), (8, 80), (16, 224), (32, 2560)):
    w, h = plane(scale)
    x = torch.randn(1, C, h, w, device=dev, requires_grad=True)
    out = rs.get_sphere_feature(x, pix, pix_sphere, scale)
    dout = torch.randn_like(out)
    out.backward(dout)                          # builds the CSR
    torch.cuda.synchronize()
    lib.scenerf_hip_profile_enable(1)
    for _ in range(reps):
        rs.get_sphere_feature(x, pix, pix_sphere, scale).backward(dout)
    torch.cuda.synchronize()
    rec = {r["name"]: r for r in _capi.profile_collect()}
    lib.scenerf_hip_profile_enable(0)
    kf = rec["sphere_resample_fwd"]; kb = rec["sphere_resample_bwd"]
    fus, bus = kf["total_ms"] * 1e3 / kf["launches"], kb["total_ms"] * 1e3 / kb["launches"]
    fbytes, bbytes = kf["bytes"] / kf["launches"], kb["bytes"] / kb["launches"]
    hip_fb = timed(lambda: rs.get_sphere_feature(x, pix, pix_sphere, scale).backward(dout), reps)
    with torch.no_grad():
        ref_f = timed(lambda: reference_form(x, scale).contiguous(), reps)
    ref_fb = timed(lambda: reference_form(x, scale).backward(dout), reps)
    rows.append(dict(level=scale, C=C, plane=[h, w], cells=out.shape[2] * out.shape[3], hip_fwd_us=fus, hip_bwd_us=bus,
                     fwd_GBps=fbytes / fus / 1e3, bwd_GBps=bbytes / bus / 1e3, hip_fwd_bwd_wall_us=hip_fb, torch_fwd_us=ref_f,
                     torch_fwd_bwd_us=ref_fb))
    tot["hip_fwd"] += fus; tot["hip_bwd"] += bus; tot["ref_fwd"] += ref_f; tot["ref_fb"] += ref_fb; tot["hip_fb"] += hip_fb
    print("level 1/%-2d C=%-4d plane %3dx%-4d: HIP fwd %6.1f us (%5.0f GB/s) bwd %6.1f us (%5.0f GB/s) | fwd+bwd wall %6.1f us | "
          "torch eager reference form: fwd %6.1f us, fwd+bwd %6.1f us" % (scale, C, h, w, fus, fbytes / fus / 1e3, bus, bbytes / bus / 1e3,
                                                                         hip_fb, ref_f, ref_fb))
# the one-time map build per level (winner scatter + pack) and CSR
rs2 = SphereResampler(out_W, out_H)
lib.scenerf_hip_profile_enable(1)
for scale in (1, 2, 4, 8, 16, 32):
    w, h = plane(scale)
    rs2.map_for(pix, pix_sphere, scale, h, w)
torch.cuda.synchronize()
mb = [r for r in _capi.profile_collect() if r["name"] == "sphere_map_build"][0]
lib.scenerf_hip_profile_enable(0)
print("six levels per image: HIP kernels fwd %.0f us + bwd %.0f us (wall incl. autograd %.0f us) vs torch eager reference form fwd %.0f us, "
      "fwd+bwd %.0f us; one-time map build %.0f us for all six levels"
      % (tot["hip_fwd"], tot["hip_bwd"], tot["hip_fb"], tot["ref_fwd"], tot["ref_fb"], mb["total_ms"] * 1e3))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "sphere_probe.json"), "w") as fh:
    json.dump(dict(levels=rows, totals=tot, map_build_us_all_levels=mb["total_ms"] * 1e3), fh, indent=1)
