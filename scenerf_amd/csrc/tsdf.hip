// TSDF fusion of one RGB-D frame into a voxel volume: the MI355X counterpart of the reference's only CUDA kernel
// (scenerf/data/utils/fusion.py:72-145, a pycuda SourceModule) -- and of the CPU path the SceneRF authors edited next to it
// (fusion.py:236-325), which is a DIFFERENT update rule.  Both are provided (`semantics`):
//   0  "gpu":  dist = min(1, (depth - z) / trunc); running weighted average of dist and of the colour channels; fp32 arithmetic,
//              pixel = roundf (half away from zero)                                                   fusion.py:96-142
//   1  "cpu":  dist = depth - z in metres; a voxel keeps the observation of smallest abs(dist) (and that observation's colour);
//              weight += obs_weight; float64 projection, pixel = round-half-even (numpy)               fusion.py:188-203,236-325
// One thread per voxel, z fastest: a wavefront touches 64 consecutive floats of each of the three volumes (coalesced); voxels that
// fall outside the frustum, on invalid depth or beyond the truncation band return before touching the volumes, as in the
// reference.  HBM-bound: 24 B per updated voxel (3 volumes read + written) plus two cached image gathers.
// Floating point: the whole library is built with -ffp-contract=off, so every operation below is rounded on its own -- the only
// compiler-independent reading of the reference kernel's source.  The golden volumes of tests/golden/tsdf_gpu_semantics.npz come from
// the reference's kernel text compiled for gfx950 the same way (oracle/build_ref.py) and the "gpu" rule reproduces them bit for bit
// (tests/test_tsdf.py); the same text under the compiler's default contraction differs in the last bit of ~1 % of the distances.
// Differences from the reference kernel, both deliberate: voxel coordinates come from exact integer division (the reference
// derives them from float(voxel_idx), which is inexact beyond 2^24 voxels), and the bound check is >= (the reference lets
// voxel_idx == n through, one element past the volumes).
#include "common.h"

struct TsdfArgs {
    float* tsdf; float* weight; float* color;
    int dx, dy, dz;
    float ox, oy, oz, voxel_size;
    float K[9];        // intrinsics, row-major
    float P[16];       // semantics 0: cam_pose (camera-to-world), row-major
    double Pinv[16];   // semantics 1: inverse of cam_pose (world-to-camera)
    double Kd[4];      // fx, fy, cx, cy as the float32 values widened (fusion.py:177-180)
    const float* color_im; const float* depth_im;
    int im_h, im_w;
    float trunc, obs_weight;
};

// what one voxel contributes: (updated?, distance, folded colour of its pixel)
template <int SEM>
__device__ static inline bool tsdf_observe(const TsdfArgs& a, int vx, int vy, int vz, float& dist, double& diff1, float& c_new) {
    int ix, iy;
    if (SEM == 0) {
        // voxel -> world (fusion.py:95-99)
        const float px = a.ox + (float)vx * a.voxel_size, py = a.oy + (float)vy * a.voxel_size, pz = a.oz + (float)vz * a.voxel_size;
        // world -> camera: R^T (p - t)   (fusion.py:100-106)
        const float tx = px - a.P[3], ty = py - a.P[7], tz = pz - a.P[11];
        const float cx = a.P[0] * tx + a.P[4] * ty + a.P[8] * tz;
        const float cy = a.P[1] * tx + a.P[5] * ty + a.P[9] * tz;
        const float cz = a.P[2] * tx + a.P[6] * ty + a.P[10] * tz;
        ix = (int)roundf(a.K[0] * (cx / cz) + a.K[2]);      // fusion.py:108-109
        iy = (int)roundf(a.K[4] * (cy / cz) + a.K[5]);
        if (ix < 0 || ix >= a.im_w || iy < 0 || iy >= a.im_h || cz < 0) return false;   // fusion.py:113-114
        const float depth = a.depth_im[(size_t)iy * a.im_w + ix];
        if (depth == 0) return false;                        // fusion.py:116-118
        const float diff = depth - cz;
        if (diff < -a.trunc) return false;                   // fusion.py:121-123
        dist = fminf(1.0f, diff / a.trunc);
    } else {
        const float px = a.ox + (float)vx * a.voxel_size, py = a.oy + (float)vy * a.voxel_size, pz = a.oz + (float)vz * a.voxel_size;
        // vox2world (fusion.py:152-162) on float32 arrays: px, py, pz above (float32 product and sum: what NumPy >= 2 computes for
        // `vol_origin[j] + vox_size * vox_coords[i, j]` with a Python-float vox_size; see oracle/tsdf_oracle.py); then
        // rigid_transform(cam_pts, inv(cam_pose)) in float64 (fusion.py:238-239, 392-397), cam2pix with round-half-even (:177-185)
        const double X = px, Y = py, Z = pz;
        const double cx = a.Pinv[0] * X + a.Pinv[1] * Y + a.Pinv[2] * Z + a.Pinv[3];
        const double cy = a.Pinv[4] * X + a.Pinv[5] * Y + a.Pinv[6] * Z + a.Pinv[7];
        const double cz = a.Pinv[8] * X + a.Pinv[9] * Y + a.Pinv[10] * Z + a.Pinv[11];
        const double fxp = rint((cx * a.Kd[0] / cz) + a.Kd[2]), fyp = rint((cy * a.Kd[1] / cz) + a.Kd[3]);
        if (!(fxp >= 0 && fxp < a.im_w && fyp >= 0 && fyp < a.im_h && cz > 0)) return false;    // fusion.py:247-251
        ix = (int)fxp; iy = (int)fyp;
        const float depth = a.depth_im[(size_t)iy * a.im_w + ix];
        const double diff = (double)depth - cz;
        if (!(depth > 0 && diff >= -(double)a.trunc)) return false;                              // fusion.py:256-258
        diff1 = diff;
        dist = (float)diff;                                                                       // dist = depth_diff (:259), stored as float32
    }
    c_new = a.color_im[(size_t)iy * a.im_w + ix];
    return true;
}

template <int SEM>
__device__ static inline void tsdf_update(const TsdfArgs& a, float dist, double diff1, float c_new, float& t, float& w, float& c) {
    const float w_old = w, w_new = w_old + a.obs_weight;
    w = w_new;                                                                  // fusion.py:126-127 / 193-194, 279
    if (SEM == 0) {
        t = (t * w_old + a.obs_weight * dist) / w_new;                          // fusion.py:128-129
        // colour: running average per channel of the folded b*65536 + g*256 + r value (fusion.py:131-141)
        const float ob = floorf(c / 65536.f), og = floorf((c - ob * 65536.f) / 256.f), orr = c - ob * 65536.f - og * 256.f;
        float nb = floorf(c_new / 65536.f), ng = floorf((c_new - nb * 65536.f) / 256.f), nr = c_new - nb * 65536.f - ng * 256.f;
        nb = fminf(roundf((ob * w_old + a.obs_weight * nb) / w_new), 255.0f);
        ng = fminf(roundf((og * w_old + a.obs_weight * ng) / w_new), 255.0f);
        nr = fminf(roundf((orr * w_old + a.obs_weight * nr) / w_new), 255.0f);
        c = nb * 65536.f + ng * 256.f + nr;
    } else if (!(fabs((double)t) < fabs(diff1))) {                              // keep the closer surface (:196-200; float64 compare)
        t = dist;
        c = c_new;                                                              // (:283-298: colour follows the mask)
    }
}

// one thread per FOUR consecutive z voxels (when dz is a multiple of 4: 16-byte accesses to the three volumes, 1 KiB per wavefront
// instruction); otherwise one voxel per thread
template <int SEM, int VPT>
__global__ __launch_bounds__(256) void tsdf_integrate_kernel(TsdfArgs a) {
    const long long n = (long long)a.dx * a.dy * a.dz;
    const long long idx = ((long long)blockIdx.x * 256 + threadIdx.x) * VPT;
    if (idx >= n) return;
    const int yz = a.dy * a.dz;
    const int vx = (int)(idx / yz);
    const int rem = (int)(idx - (long long)vx * yz);
    const int vy = rem / a.dz, vz = rem - vy * a.dz;
    float dist[VPT], cn[VPT];
    double d1[VPT];
    bool on[VPT], any = false;
#pragma unroll
    for (int e = 0; e < VPT; ++e) {
        dist[e] = 0.f; cn[e] = 0.f; d1[e] = 0.0;
        on[e] = tsdf_observe<SEM>(a, vx, vy, vz + e, dist[e], d1[e], cn[e]);
        any |= on[e];
    }
    if (!any) return;   // outside the frustum / invalid depth / beyond the truncation band: the volumes are not touched
    if (VPT == 4) {
        float4 t = *(const float4*)(a.tsdf + idx), w = *(const float4*)(a.weight + idx), c = *(const float4*)(a.color + idx);
        float* tp = &t.x; float* wp = &w.x; float* cp = &c.x;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (on[e]) tsdf_update<SEM>(a, dist[e], d1[e], cn[e], tp[e], wp[e], cp[e]);
        *(float4*)(a.tsdf + idx) = t; *(float4*)(a.weight + idx) = w; *(float4*)(a.color + idx) = c;
    } else {
        float t = a.tsdf[idx], w = a.weight[idx], c = a.color[idx];
        tsdf_update<SEM>(a, dist[0], d1[0], cn[0], t, w, c);
        a.tsdf[idx] = t; a.weight[idx] = w; a.color[idx] = c;
    }
}

extern "C" int scenerf_hip_tsdf_integrate(float* tsdf_vol, float* weight_vol, float* color_vol, const int32_t vol_dim[3],
                                          const float vol_origin[3], double voxel_size, const float cam_intr[9], const float cam_pose[16],
                                          const double cam_pose_inv[16], const float* color_im, const float* depth_im, int im_h, int im_w,
                                          float trunc_margin, float obs_weight, int semantics, scenerf_stream_t stream) {
    SRF_CHECK(tsdf_vol && weight_vol && color_vol && vol_dim && vol_origin && cam_intr && cam_pose && color_im && depth_im,
              "tsdf_integrate: NULL argument");
    SRF_CHECK(vol_dim[0] > 0 && vol_dim[1] > 0 && vol_dim[2] > 0 && im_h > 0 && im_w > 0 && voxel_size > 0, "tsdf_integrate: bad sizes");
    SRF_CHECK(semantics == 0 || (semantics == 1 && cam_pose_inv), "tsdf_integrate: semantics must be 0 (gpu) or 1 (cpu, needs cam_pose_inv)");
    TsdfArgs a;
    a.tsdf = tsdf_vol; a.weight = weight_vol; a.color = color_vol;
    a.dx = vol_dim[0]; a.dy = vol_dim[1]; a.dz = vol_dim[2];
    a.ox = vol_origin[0]; a.oy = vol_origin[1]; a.oz = vol_origin[2]; a.voxel_size = (float)voxel_size;
    for (int i = 0; i < 9; ++i) a.K[i] = cam_intr[i];
    for (int i = 0; i < 16; ++i) { a.P[i] = cam_pose[i]; a.Pinv[i] = cam_pose_inv ? cam_pose_inv[i] : 0.0; }
    a.Kd[0] = cam_intr[0]; a.Kd[1] = cam_intr[4]; a.Kd[2] = cam_intr[2]; a.Kd[3] = cam_intr[5];
    a.color_im = color_im; a.depth_im = depth_im; a.im_h = im_h; a.im_w = im_w;
    a.trunc = trunc_margin; a.obs_weight = obs_weight;
    const long long n = (long long)a.dx * a.dy * a.dz;
    SRF_CHECK(n < (1ll << 39), "tsdf_integrate: volume too large");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "tsdf_integrate", 0, (double)n * 24.0);
    if (a.dz % 4 == 0) {
        const unsigned grid = (unsigned)((n / 4 + 255) / 256);
        if (semantics == 0) tsdf_integrate_kernel<0, 4><<<grid, 256, 0, s>>>(a);
        else tsdf_integrate_kernel<1, 4><<<grid, 256, 0, s>>>(a);
    } else {
        const unsigned grid = (unsigned)((n + 255) / 256);
        if (semantics == 0) tsdf_integrate_kernel<0, 1><<<grid, 256, 0, s>>>(a);
        else tsdf_integrate_kernel<1, 1><<<grid, 256, 0, s>>>(a);
    }
    SRF_LAUNCH_CHECK("tsdf_integrate_kernel");
    return 0;
}
