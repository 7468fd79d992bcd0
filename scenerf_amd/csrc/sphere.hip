// Image -> sphere resampling of the encoder levels: the MI355X counterpart of DecoderSphere.get_sphere_feature
// (reference scenerf/models/unet2d_sphere.py:138-165), which the reference runs six times per image as
//   zeros(out_W, out_H, 2) - 10  ->  index_put (451 k pixels, duplicate cells)  ->  normalise  ->  F.grid_sample(bilinear)  ->  permute.
// Here the scatter happens ONCE per (intrinsics, level) and is deterministic: the map depends only on the pixel grid, so the
// host caches `src` (per sphere cell: the packed integer coordinates (sy << 16 | sx) of the winning pixel's `pix // scale`, or -1)
// and each call is one gather kernel.  Duplicates: the LAST pixel in index order wins -- what the reference's index_put does on one
// CPU thread; on a GPU its winner is undefined (SURVEY §2a "non-deterministic scatter").
//   forward   one thread per sphere cell x 4 planes: the four taps and fp32 weights are derived from (sx, sy) with the SAME
//             operation sequence as the reference's normalise + ATen's grid_sampler_unnormalize (the sample point is sx - 0.5 only
//             up to fp32 rounding, weights 0.5 +- 1e-4 at W = 1220), so results agree with the oracle bit for bit;
//             output written directly as (B, C, out_H, out_W) -- the reference returns a permuted view the next conv re-packs.
//   backward  gather form, no atomics: one thread per SOURCE pixel x 4 planes walks the cells that reference its four neighbours
//             through a CSR (cells grouped by source pixel on a (H+1) x (W+1) grid, built once with the map); every dx element is written exactly once
//             (no zero fill) and the summation order is fixed.
// Both are HBM/L2 gathers: algorithmic bytes = planes * (H*W + out_H*out_W) * 4 + 4 per cell of map.
#include "common.h"

static constexpr int SPH_CG = 4;        // planes per thread
static constexpr int SPH_THREADS = 256;

__global__ void sphere_winner_kernel(const long long* __restrict__ pix_sphere, long long n_pix, float scale, int out_w, int out_h,
                                     int* __restrict__ winner) {
    const long long p = (long long)blockIdx.x * SPH_THREADS + threadIdx.x;
    if (p >= n_pix) return;
    // torch.round(pix_sphere / scale).long() then clamp (unet2d_sphere.py:141-144); rintf = half to even like torch.round
    int u = (int)rintf((float)pix_sphere[2 * p] / scale);
    int v = (int)rintf((float)pix_sphere[2 * p + 1] / scale);
    u = min(max(u, 0), out_w - 1);
    v = min(max(v, 0), out_h - 1);
    atomicMax(&winner[v * out_w + u], (int)p);
}

__global__ void sphere_src_kernel(const float* __restrict__ pix, const int* __restrict__ winner, float scale, int ncell,
                                  int* __restrict__ src) {
    const int cell = blockIdx.x * SPH_THREADS + threadIdx.x;
    if (cell >= ncell) return;
    const int w = winner[cell];
    int s = -1;
    if (w >= 0) {   // pix // scale (unet2d_sphere.py:142)
        const int sx = (int)floorf(pix[2 * (long long)w] / scale), sy = (int)floorf(pix[2 * (long long)w + 1] / scale);
        s = (sy << 16) | (sx & 0xFFFF);
    }
    src[cell] = s;
}

// map value -> sample coordinate, op by op (unet2d_sphere.py:151-153, then ((g + 1) * size - 1) / 2); the library is compiled
// with -ffp-contract=off, so every operation rounds on its own like the eager reference.
__device__ static inline float sphere_coord(int s, float size) {
    float g = (float)s / size;
    g = g * 2.0f;
    g = g - 1.0f;
    float t = g + 1.0f;
    t = t * size;
    t = t - 1.0f;
    return t / 2.0f;
}

// element (plane, cell) of the sphere-side tensor: planar (B*C, out_h, out_w) when out_C == 0, channels-last (B, out_h, out_w, out_C)
// otherwise (plane = b * out_C + c)
__device__ static inline long long sph_idx(long long plane, int cell, int ncell, int out_C) {
    return out_C ? ((plane / out_C) * ncell + cell) * out_C + plane % out_C : plane * ncell + cell;
}

__global__ void __launch_bounds__(SPH_THREADS)
sphere_fwd_kernel(const float* __restrict__ x, long long planes, int H, int W, const int* __restrict__ src, int ncell, int out_C,
                  float* __restrict__ out) {
    const int cell = blockIdx.x * SPH_THREADS + threadIdx.x;
    if (cell >= ncell) return;
    const long long c0 = (long long)blockIdx.y * SPH_CG;
    const int s = src[cell];
    const long long hw = (long long)H * W;
    if (s < 0) {
#pragma unroll
        for (int j = 0; j < SPH_CG; ++j)
            if (c0 + j < planes) out[sph_idx(c0 + j, cell, ncell, out_C)] = 0.0f;
        return;
    }
    const float ix = sphere_coord(s & 0xFFFF, (float)W), iy = sphere_coord(s >> 16, (float)H);
    const float fx0 = floorf(ix), fy0 = floorf(iy);
    const float fx1 = fx0 + 1.0f, fy1 = fy0 + 1.0f;
    const float w_nw = (fx1 - ix) * (fy1 - iy), w_ne = (ix - fx0) * (fy1 - iy), w_sw = (fx1 - ix) * (iy - fy0), w_se = (ix - fx0) * (iy - fy0);
    const int x0 = (int)fx0, y0 = (int)fy0;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x0 + 1 >= 0 && x0 + 1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y0 + 1 >= 0 && y0 + 1 < H;
    const long long o_nw = (long long)y0 * W + x0;
#pragma unroll
    for (int j = 0; j < SPH_CG; ++j) {
        if (c0 + j >= planes) break;
        const float* xp = x + (c0 + j) * hw;
        const float a = (vx0 && vy0) ? xp[o_nw] : 0.0f;
        const float b = (vx1 && vy0) ? xp[o_nw + 1] : 0.0f;
        const float c = (vx0 && vy1) ? xp[o_nw + W] : 0.0f;
        const float d = (vx1 && vy1) ? xp[o_nw + W + 1] : 0.0f;
        float acc = a * w_nw;
        acc = acc + b * w_ne;
        acc = acc + c * w_sw;
        acc = acc + d * w_se;
        out[sph_idx(c0 + j, cell, ncell, out_C)] = acc;
    }
}

__global__ void __launch_bounds__(SPH_THREADS)
sphere_bwd_kernel(const float* __restrict__ dout, long long planes, int H, int W, const int* __restrict__ row_ptr,
                  const int* __restrict__ cells, int ncell, int out_C, float* __restrict__ dx) {
    const long long p = (long long)blockIdx.x * SPH_THREADS + threadIdx.x;
    const long long hw = (long long)H * W;
    if (p >= hw) return;
    const long long c0 = (long long)blockIdx.y * SPH_CG;
    const int py = (int)(p / W), px = (int)(p % W);
    float acc[SPH_CG];
#pragma unroll
    for (int j = 0; j < SPH_CG; ++j) acc[j] = 0.0f;
    // pixel (px, py) is a tap of the cells whose source pixel is (px + ex, py + ey), ex, ey in {0, 1}
    // (the CSR is indexed on a (H+1) x (W+1) grid: a map entry one past the plane still has in-range taps)
    for (int ey = 0; ey < 2; ++ey) {
        const int qy = py + ey;
        for (int ex = 0; ex < 2; ++ex) {
            const int qx = px + ex;
            const long long q = (long long)qy * (W + 1) + qx;
            const int e0 = row_ptr[q], e1 = row_ptr[q + 1];
            if (e0 == e1) continue;
            const float ix = sphere_coord(qx, (float)W), iy = sphere_coord(qy, (float)H);
            const float fx0 = floorf(ix), fy0 = floorf(iy);
            const int tx = px - (int)fx0, ty = py - (int)fy0;
            if ((tx | ty) & ~1) continue;      // not one of this cell group's four taps
            const float wx = tx ? (ix - fx0) : (fx0 + 1.0f - ix);
            const float wy = ty ? (iy - fy0) : (fy0 + 1.0f - iy);
            const float w = wx * wy;
            for (int e = e0; e < e1; ++e) {
                const int cell = cells[e];
#pragma unroll
                for (int j = 0; j < SPH_CG; ++j)
                    if (c0 + j < planes) acc[j] = acc[j] + dout[sph_idx(c0 + j, cell, ncell, out_C)] * w;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < SPH_CG; ++j)
        if (c0 + j < planes) dx[(c0 + j) * hw + p] = acc[j];
}

extern "C" int scenerf_hip_sphere_map_build(const float* pix, const int64_t* pix_sphere, int64_t n_pix, int scale, int out_w, int out_h,
                                            int32_t* winner, int32_t* src, scenerf_stream_t stream) {
    SRF_CHECK(pix && pix_sphere && winner && src, "sphere_map_build: NULL argument");
    SRF_CHECK(n_pix > 0 && n_pix < (1ll << 31) && scale >= 1 && out_w > 0 && out_h > 0 && (long long)out_w * out_h < (1ll << 31),
              "sphere_map_build: bad sizes");
    hipStream_t s = as_stream(stream);
    const int ncell = out_w * out_h;
    SrfLaunchScope ps(s, "sphere_map_build", 0, (double)n_pix * 24.0 + (double)ncell * 8.0);
    SRF_HIP(hipMemsetAsync(winner, 0xFF, (size_t)ncell * sizeof(int32_t), s));
    sphere_winner_kernel<<<(unsigned)((n_pix + SPH_THREADS - 1) / SPH_THREADS), SPH_THREADS, 0, s>>>(
        (const long long*)pix_sphere, (long long)n_pix, (float)scale, out_w, out_h, winner);
    sphere_src_kernel<<<(unsigned)((ncell + SPH_THREADS - 1) / SPH_THREADS), SPH_THREADS, 0, s>>>(pix, winner, (float)scale, ncell, src);
    SRF_LAUNCH_CHECK("sphere_map_build");
    return 0;
}

static bool sphere_dims_ok(int64_t planes, int H, int W, int out_w, int out_h) {
    return planes > 0 && (planes + SPH_CG - 1) / SPH_CG <= 65535 && H > 0 && W > 0 && H < 32768 && W < 65536 && out_w > 0 && out_h > 0 &&
           (long long)out_w * out_h < (1ll << 31) && (long long)H * W < (1ll << 31);
}

static int sphere_forward(const float* x, int64_t planes, int H, int W, const int32_t* src, int out_w, int out_h, int out_C, float* out,
                          scenerf_stream_t stream) {
    SRF_CHECK(x && src && out, "sphere_resample_forward: NULL argument");
    SRF_CHECK(out_C >= 0 && (out_C == 0 || planes % out_C == 0), "sphere_resample_forward: planes must be a multiple of the channel count");
    SRF_CHECK(sphere_dims_ok(planes, H, W, out_w, out_h), "sphere_resample_forward: bad sizes");
    hipStream_t s = as_stream(stream);
    const int ncell = out_w * out_h;
    SrfLaunchScope ps(s, "sphere_resample_fwd", 0, (double)planes * ((double)H * W + ncell) * 4.0 + ncell * 4.0);
    dim3 grid((unsigned)((ncell + SPH_THREADS - 1) / SPH_THREADS), (unsigned)((planes + SPH_CG - 1) / SPH_CG));
    sphere_fwd_kernel<<<grid, SPH_THREADS, 0, s>>>(x, (long long)planes, H, W, src, ncell, out_C, out);
    SRF_LAUNCH_CHECK("sphere_resample_forward");
    return 0;
}

static int sphere_backward(const float* dout, int64_t planes, int H, int W, const int32_t* row_ptr, const int32_t* cells, int out_w, int out_h,
                           int out_C, float* dx, scenerf_stream_t stream) {
    SRF_CHECK(dout && row_ptr && cells && dx, "sphere_resample_backward: NULL argument");
    SRF_CHECK(out_C >= 0 && (out_C == 0 || planes % out_C == 0), "sphere_resample_backward: planes must be a multiple of the channel count");
    SRF_CHECK(sphere_dims_ok(planes, H, W, out_w, out_h), "sphere_resample_backward: bad sizes");
    hipStream_t s = as_stream(stream);
    const int ncell = out_w * out_h;
    const long long hw = (long long)H * W;
    SrfLaunchScope ps(s, "sphere_resample_bwd", 0, (double)planes * ((double)hw + ncell) * 4.0 + hw * 4.0);
    dim3 grid((unsigned)((hw + SPH_THREADS - 1) / SPH_THREADS), (unsigned)((planes + SPH_CG - 1) / SPH_CG));
    sphere_bwd_kernel<<<grid, SPH_THREADS, 0, s>>>(dout, (long long)planes, H, W, row_ptr, cells, ncell, out_C, dx);
    SRF_LAUNCH_CHECK("sphere_resample_backward");
    return 0;
}

extern "C" int scenerf_hip_sphere_resample_forward(const float* x, int64_t planes, int H, int W, const int32_t* src, int out_w, int out_h,
                                                   float* out, scenerf_stream_t stream) {
    return sphere_forward(x, planes, H, W, src, out_w, out_h, 0, out, stream);
}
extern "C" int scenerf_hip_sphere_resample_backward(const float* dout, int64_t planes, int H, int W, const int32_t* row_ptr,
                                                    const int32_t* cells, int out_w, int out_h, float* dx, scenerf_stream_t stream) {
    return sphere_backward(dout, planes, H, W, row_ptr, cells, out_w, out_h, 0, dx, stream);
}
// channels-last sphere side: out / dout are (B, out_h, out_w, C) with planes = B * C -- the layout the renderer reads in place
// (scenerf_cfg.map_chw == 2), so that no conversion sits between the decoder and the gather
extern "C" int scenerf_hip_sphere_resample_forward_nhwc(const float* x, int64_t planes, int C, int H, int W, const int32_t* src, int out_w,
                                                        int out_h, float* out, scenerf_stream_t stream) {
    SRF_CHECK(C > 0, "sphere_resample_forward_nhwc: C must be positive");
    return sphere_forward(x, planes, H, W, src, out_w, out_h, C, out, stream);
}
extern "C" int scenerf_hip_sphere_resample_backward_nhwc(const float* dout, int64_t planes, int C, int H, int W, const int32_t* row_ptr,
                                                         const int32_t* cells, int out_w, int out_h, float* dx, scenerf_stream_t stream) {
    SRF_CHECK(C > 0, "sphere_resample_backward_nhwc: C must be positive");
    return sphere_backward(dout, planes, H, W, row_ptr, cells, out_w, out_h, C, dx, stream);
}
