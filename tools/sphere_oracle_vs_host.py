"""CPU only: the oracle's geometry intermediates (torch ops on THIS host) against csrc/sphere_exact.h built with gcc, at a parity_full case's
size -- finds which torch op leaves the restated sequence on a given host CPU.  usage: sphere_oracle_vs_host.py [case]"""
import ctypes as C, dataclasses, os, subprocess, sys, tempfile
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenerf_oracle as orc
from scenerf_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "kitti_c2_r1200_n128"
CASES = {"kitti_c2_r1200_n128": dict(variant="kitti", R=1200, U=64, P=16, sphere=(1500, 452), img=(1220, 370), pose=(1.0, 0.0), seed=900),
         "bf_c4_r1080_n96": dict(variant="bf", R=1080, U=64, P=8, sphere=(960, 720), img=(640, 480), pose=(0.3, 8.0), seed=910)}
spec = CASES[name]
sd = spec["seed"]
pix = synth.stride2_pixels(spec["img"], spec["R"], sd + 4)
nu, ng = synth.sampling_noise(spec["R"], spec["U"], 4 * spec["P"], sd + 5)
K = (synth.kitti_cam_K() if spec["variant"] == "kitti" else synth.bundlefusion_cam_K()).contiguous()
T = synth.rel_pose(*spec["pose"]).contiguous()
mk = orc.OracleConfig.kitti if spec["variant"] == "kitti" else orc.OracleConfig.bundlefusion
cfg = mk(n_pts_uni=spec["U"], n_pts_per_gaussian=spec["P"], index_rule="pinned")
iK = torch.inverse(K).contiguous().clone()
R, U = spec["R"], spec["U"]
if os.environ.get("SRF_THREADS"):
    torch.set_num_threads(int(os.environ["SRF_THREADS"]))
print("torch", torch.__version__, "capability", torch.backends.cpu.get_cpu_capability(), "threads", torch.get_num_threads())
td = tempfile.mkdtemp(); so = os.path.join(td, "libsx.so")
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "csrc", "sphere_exact_host.c"), "-lm"])
host = C.CDLL(so); vp = C.c_void_p
host.srf_host_points_to_sphere.argtypes = [vp, C.c_size_t, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
host.srf_host_rays.argtypes = [vp, C.c_size_t, vp, vp, vp, vp]
host.srf_host_sample_points.argtypes = [vp, vp, C.c_size_t, C.c_size_t, vp, vp]
ne = lambda a, b: int((a.contiguous().view(torch.int32) != b.contiguous().view(torch.int32)).sum())
dirs, unit = orc.ray_directions(pix, iK)
vd = (T[:3, :3] @ dirs.T).T
hu, hv = torch.empty(R, 3), torch.empty(R, 3)
pixc = pix.contiguous()
host.srf_host_rays(pixc.data_ptr(), R, iK.data_ptr(), T.data_ptr(), hu.data_ptr(), hv.data_ptr())
print("unit:", ne(hu, unit), "viewdir:", ne(hv, vd))
g = torch.Generator().manual_seed(1)
S = 132
dist = (torch.rand(R, S, generator=g) * cfg.max_sample_depth).contiguous()
pts = orc.to_frame((dist.unsqueeze(-1) * unit.reshape(R, 1, 3)).reshape(-1, 3), T)
hp = torch.empty(R * S, 3)
host.srf_host_sample_points(hu.data_ptr(), dist.data_ptr(), R, S, T.data_ptr(), hp.data_ptr())
print("points:", ne(hp, pts), "of", 3 * R * S)
opix = orc.project_to_pixels(pts, K)
idx, coords = orc.sphere_coords(opix, iK, cfg, return_float=True)
M = R * S
hi, hc, hx = torch.empty(M, 2, dtype=torch.int32), torch.empty(M, 2), torch.empty(M, 2)
consts = torch.tensor(cfg.fov, dtype=torch.float32)
ptsc = pts.contiguous()
host.srf_host_points_to_sphere(ptsc.data_ptr(), M, K.data_ptr(), iK.data_ptr(), consts.data_ptr(), cfg.sphere_W, cfg.sphere_H, hi.data_ptr(), hc.data_ptr(), hx.data_ptr())
print("pix:", ne(hx, opix), "coords x:", ne(hc[:, 0], coords[:, 0]), "coords y:", ne(hc[:, 1], coords[:, 1]), "idx:", int((hi.long() != idx).any(1).sum()))
bad = (hx.view(torch.int32) != opix.contiguous().view(torch.int32)).any(1)
if bool(bad.any()):
    j = bad.nonzero()[:, 0]
    print("pix mismatch rows:", j[:10].tolist(), "(M = %d; M mod 16 = %d)" % (M, M % 16))
bad = (hc.view(torch.int32) != coords.contiguous().view(torch.int32))
if bool(bad.any()):
    print("coords mismatch rows:", bad.any(1).nonzero()[:10, 0].tolist())
if len(sys.argv) > 2:
    h = (K @ pts.T).T
    print("rows 0..3 of K @ p (torch):", [[x.hex() for x in r] for r in h[:4].tolist()])
    def fma(a, b, c):
        import numpy as np
        return float(np.float32(np.float64(a) * np.float64(b) + np.float64(c)))
    import numpy as np
    for r in range(4):
        p = pts[r].tolist(); Kl = K.tolist()
        seq = [float(np.float32(np.float32(np.float32(Kl[i][0] * np.float32(p[0])) + np.float32(np.float32(Kl[i][1]) * np.float32(p[1]))) + np.float32(np.float32(Kl[i][2]) * np.float32(p[2])))) for i in range(3)]
        f = [fma(Kl[i][2], p[2], fma(Kl[i][1], p[1], float(np.float32(np.float32(Kl[i][0]) * np.float32(p[0]))))) for i in range(3)]
        f_rev = [fma(Kl[i][0], p[0], fma(Kl[i][1], p[1], float(np.float32(np.float32(Kl[i][2]) * np.float32(p[2]))))) for i in range(3)]
        print(r, "k-ordered fma", [x.hex() for x in f], "| reversed fma", [x.hex() for x in f_rev], "| no-fma seq", [x.hex() for x in seq])
    print("h storage offset / strides:", h.storage_offset(), h.stride(), "pts strides", pts.stride(), pts.storage_offset(), "pts data_ptr % 64:", pts.data_ptr() % 64)
if len(sys.argv) > 2:
    import numpy as np
    print("opix rows 0..3:", [[x.hex() for x in r] for r in opix[:4].tolist()])
    print("host  rows 0..3:", [[x.hex() for x in r] for r in hx[:4].tolist()])
    print("exact div      :", [[float(np.float32(h[r, c].item()) / np.float32(h[r, 2].item())).hex() for c in range(2)] for r in range(4)])
    q = h[:, :2] / h[:, 2:3]
    print("torch div rows :", [[x.hex() for x in r] for r in q[:4].tolist()])
    q2 = h[:, :2].contiguous() / h[:, 2:3].contiguous()
    print("torch div contiguous rows :", [[x.hex() for x in r] for r in q2[:4].tolist()], "differs from strided on", int((q2 != q).any(1).sum()), "rows")
