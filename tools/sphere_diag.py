"""Where does the device leave the host's operation sequence?  ray_setup + encode_points on the GPU against csrc/sphere_exact.h built for the
host with gcc, stage by stage, on random rays at the KITTI geometry.  usage: sphere_diag.py [rays] [samples]"""
import ctypes as C, os, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from scenerf_amd import _capi, synth
from scenerf_amd.config import RenderConfig
import sleef_acos

R = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
td = tempfile.mkdtemp()
so = os.path.join(td, "libsx.so")
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "csrc", "sphere_exact_host.c"), "-lm"])
host = C.CDLL(so)
vp = C.c_void_p
host.srf_host_points_to_sphere.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
host.srf_host_rays.argtypes = [vp, C.c_size_t, vp, vp, vp, vp]
host.srf_host_sample_points.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, vp]
lib = _capi.load()
dev = "cuda"
BF = os.environ.get("SRF_VARIANT", "kitti") == "bf"
rcfg = (RenderConfig.bundlefusion if BF else RenderConfig.kitti)(precision="fp32", n_pts_uni=S, n_pts_per_gaussian=16)
cc = rcfg.to_c()
g = torch.Generator().manual_seed(5)
K = (synth.bundlefusion_cam_K() if BF else synth.kitti_cam_K()).contiguous(); iK = torch.inverse(K).contiguous().clone()
T = (synth.rel_pose(0.3, 8.0) if BF else synth.rel_pose(1.0, 0.0)).contiguous()
IW, IH, D = (640, 480, 12.0) if BF else (1220, 370, 100.0)
pix = torch.stack([torch.randint(0, IW, (R,), generator=g).float(), torch.randint(0, IH, (R,), generator=g).float()], 1).contiguous()
dist = (torch.rand(R, S, generator=g) * D).contiguous()
st = torch.cuda.current_stream().cuda_stream
unit = torch.empty((R, 3), device=dev); vd = torch.empty((R, 3), device=dev); du = torch.empty((R, S), device=dev)
lin = torch.linspace(0.2, D, steps=S).to(dev); nu = torch.zeros(R, S, device=dev)
pd, ikd, Td, Kd = pix.to(dev), iK.to(dev), T.to(dev), K.to(dev)
_capi.check(lib.scenerf_hip_ray_setup(C.byref(cc), pd.data_ptr(), ikd.data_ptr(), Td.data_ptr(), lin.data_ptr(), nu.data_ptr(), None, R, unit.data_ptr(), vd.data_ptr(), du.data_ptr(), st), "ray_setup")
hu, hv = torch.empty(R, 3), torch.empty(R, 3)
host.srf_host_rays(pix.data_ptr(), R, iK.data_ptr(), T.data_ptr(), hu.data_ptr(), hv.data_ptr())
ne = lambda a, b: int((a.contiguous().view(torch.int32) != b.contiguous().view(torch.int32)).sum())
print("unit_dir elements differing:", ne(unit.cpu(), hu), " viewdir:", ne(vd.cpu(), hv), "of", 3 * R)
M = R * S
pts = torch.empty((M, 3), device=dev); idx = torch.empty((M, 2), dtype=torch.int32, device=dev); xenc = torch.empty((M, 48), device=dev)
dd = dist.to(dev); hud = hu.to(dev); hvd = hv.to(dev)
_capi.check(lib.scenerf_hip_encode_points(C.byref(cc), dd.data_ptr(), S, S, hud.data_ptr(), hvd.data_ptr(), Kd.data_ptr(), ikd.data_ptr(), Td.data_ptr(), M,
                                          pts.data_ptr(), idx.data_ptr(), xenc.data_ptr(), None, st), "encode")
hp = torch.empty(M, 3)
host.srf_host_sample_points(hu.data_ptr(), dist.data_ptr(), R, S, T.data_ptr(), hp.data_ptr())
print("sample point elements differing:", ne(pts.cpu(), hp), "of", 3 * M)
hi, hc, hx = torch.empty(M, 2, dtype=torch.int32), torch.empty(M, 2), torch.empty(M, 2)
consts = torch.tensor([cc.v_min, cc.v_fov, cc.h_min, cc.h_fov], dtype=torch.float32)
host.srf_host_points_to_sphere(hp.data_ptr(), M, K.data_ptr(), iK.data_ptr(), consts.data_ptr(), cc.sphere_W, cc.sphere_H, hi.data_ptr(), hc.data_ptr(), hx.data_ptr())
d = idx.cpu() != hi
print("sphere idx differing: columns %d rows %d of %d" % (int(d[:, 0].sum()), int(d[:, 1].sum()), M))
# the two routines alone on the chain's own arguments
c = (iK @ torch.cat([hx, torch.ones(M, 1)], 1).T).T.contiguous()
n = torch.linalg.norm(c, ord=2, dim=1)
arg = (-c[:, 1] / n).contiguous()
out = torch.empty(M, device=dev); out2 = torch.empty(M, device=dev)
ad, zd, xd = arg.to(dev), c[:, 2].contiguous().to(dev), c[:, 0].contiguous().to(dev)
_capi.check(lib.scenerf_hip_test_acos_atan2(ad.data_ptr(), None, M, out.data_ptr(), None, st), "acos")
_capi.check(lib.scenerf_hip_test_acos_atan2(zd.data_ptr(), xd.data_ptr(), M, None, out2.data_ptr(), st), "atan2")
print("device acos vs torch's SLEEF on the chain's arguments:", ne(out.cpu(), sleef_acos.acos(arg)), " atan2:", ne(out2.cpu(), sleef_acos.atan2(c[:, 2].contiguous(), c[:, 0].contiguous())))
if bool(d.any()):
    i = d.any(1).nonzero()[:5, 0]
    for j in i.tolist():
        print("sample", j, "pt", [x.hex() for x in hp[j].tolist()], "pix", hx[j].tolist(), "host coords", [x.hex() for x in hc[j].tolist()], "host idx", hi[j].tolist(), "gpu idx", idx[j].cpu().tolist(),
              "acos arg", arg[j].item().hex(), "dev acos", out[j].item().hex(), "sleef", sleef_acos.acos(arg[j:j+1])[0].item().hex())
