// Loss-side gathers of a training step for gfx950, fused with the renderer's per-ray outputs: for every ray the
// source colour (sample_pix_features, reference scenerf/models/utils.py:250-266), |rendered colour - source colour|
// (scenerf.py:302-307) and the photometric reprojection term of scenerf.py:349-386 -- back-project the pixel at the rendered depth,
// move it into the target frame, project (utils.py:298-315), sample the target image there and at the source pixel, min(L1, L1 of the
// identity reprojection + noise) -- plus its masked mean over the rays whose target point lies in front of the camera.  One thread per
// ray does all of it in registers (three bilinear samples, two 4x4 transforms); the derivative of a ray's term w.r.t. its rendered
// depth is closed-form (through the projection and the bilinear weights) and is produced by the same pass, so the backward is one
// multiply per ray.  The reference spends ~25 eager kernels and two boolean-index host syncs on this.
#include "common.h"

struct LossArgs {
    const float* pix;        // [R][2] source pixels
    const float* color;      // [R][3] rendered colour
    const float* depth;      // [R] rendered depth
    const float* img_s;      // [3][H][W]
    const float* img_t;      // [3][H][W]
    const float* noise;      // [R] or NULL: noise[r] * noise_scale is added to the identity term (the reference draws randn * 1e-5 there)
    float noise_scale;
    const unsigned long long* rng;   // or NULL; noise == NULL: {seed, calls so far}: the noise is made in the kernel (source_loss only)
    const float *K, *invK, *T;    // device: row-major 3x3, 3x3, and the 4x4 source->target transform (top three rows are read)
    int R, H, W;
    float* loss_color;       // [R][3]
    float* ray_term;         // [R] min(reprojection, identity)
    float* valid;            // [R] 1 if the target point is in front of the camera
    float* dterm_ddepth;     // [R] derivative of ray_term w.r.t. depth (0 where the identity term wins or the projection is clamped)
    float* col_src;          // [R][3] (kept: the colour loss's sign in the backward)
    float* loss_rep;         // [1] masked mean of ray_term (atomically accumulated numerator / denominator in acc[0..1])
    float* acc;              // [2] zeroed by the caller
};

// bilinear sample of one channel plane at pixel coords (px, py), grid_sample(align_corners=False, padding zeros) of utils.py:250-266:
// x = px / (W - 1) * W - 0.5.  Returns the value and its derivatives w.r.t. px, py.
__device__ static inline void bilin3(const float* img, int H, int W, float px, float py, float (&v)[3], float (&dx)[3], float (&dy)[3]) {
    const float sx = (float)W / (float)(W - 1), sy = (float)H / (float)(H - 1);
    // the reference's operation order: g = (p / (W - 1) - 0.5) * 2 ; x = ((g + 1) * W - 1) / 2
    const float gx = (px / (float)(W - 1) - 0.5f) * 2.f, gy = (py / (float)(H - 1) - 0.5f) * 2.f;
    const float x = ((gx + 1.f) * (float)W - 1.f) * 0.5f, y = ((gy + 1.f) * (float)H - 1.f) * 0.5f;
    const float x0f = floorf(x), y0f = floorf(y);
    const int x0 = (int)x0f, y0 = (int)y0f;
    const float fx = x - x0f, fy = y - y0f;
    const bool okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W, oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
    const size_t plane = (size_t)H * W;
    // branch-free taps: an out-of-range tap reads a clamped (valid) texel and is replaced by zero afterwards, so that the twelve loads of
    // a sample -- 36 per ray -- are all in flight together.  Guarded loads compile to one basic block and one s_waitcnt each: 36
    // dependent memory latencies per thread, 83 us for 1,200 rays (r04_e; the images are 5 MB each, every tap a miss)
    const int xa = min(max(x0, 0), W - 1), xb = min(max(x0 + 1, 0), W - 1), ya = min(max(y0, 0), H - 1), yb = min(max(y0 + 1, 0), H - 1);
    float tv[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float* p = img + c * plane;
        tv[c][0] = p[(size_t)ya * W + xa]; tv[c][1] = p[(size_t)ya * W + xb];
        tv[c][2] = p[(size_t)yb * W + xa]; tv[c][3] = p[(size_t)yb * W + xb];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float a = (okx0 && oky0) ? tv[c][0] : 0.f, b = (okx1 && oky0) ? tv[c][1] : 0.f;
        const float d = (okx0 && oky1) ? tv[c][2] : 0.f, e = (okx1 && oky1) ? tv[c][3] : 0.f;
        v[c] = a * (1.f - fx) * (1.f - fy) + b * fx * (1.f - fy) + d * (1.f - fx) * fy + e * fx * fy;
        dx[c] = ((b - a) * (1.f - fy) + (e - d) * fy) * sx;
        dy[c] = ((d - a) * (1.f - fx) + (e - b) * fx) * sy;
    }
}

// ---- N(0, 1) noise made in the kernel (common.h: Philox4x32-10 keyed by a seed, counted by (ray, call), Box-Muller): the reference draws
// torch.randn(R) * 1e-5 per source frame to break ties between the two reprojection terms (scenerf.py:378) -- as a torch call that is two
// more launches (29 us in a step's trace, r04_f) in front of this kernel
__device__ static inline float sl_normal(const unsigned long long* state, int r) {
    uint32_t o[4];
    srf_philox((uint32_t)r, (uint32_t)state[1], 0x10551u, state[0], o);
    return srf_normal(o[0], o[1]);
}

// one ray of the colour + reprojection terms: writes the per-ray records and returns (term, valid, sum_c |colour - source colour|)
__device__ static inline void loss_ray(const LossArgs& p, int r, float& term, float& val, float& col_l1) {
    const float px = p.pix[2 * r], py = p.pix[2 * r + 1], depth = p.depth[r];
    float cs[3], ci[3], ct[3], d0[3], d1[3], tx[3], ty[3];
    bilin3(p.img_s, p.H, p.W, px, py, cs, d0, d1);
    bilin3(p.img_t, p.H, p.W, px, py, ci, d0, d1);
    // back-project, transform, project (scenerf.py:355-368)
    const float vx = p.invK[0] * px + p.invK[1] * py + p.invK[2], vy = p.invK[3] * px + p.invK[4] * py + p.invK[5],
                vz = p.invK[6] * px + p.invK[7] * py + p.invK[8];
    const float sxp = depth * vx, syp = depth * vy, szp = depth * vz;
    const float cx = p.T[0] * sxp + p.T[1] * syp + p.T[2] * szp + p.T[3], cy = p.T[4] * sxp + p.T[5] * syp + p.T[6] * szp + p.T[7],
                cz = p.T[8] * sxp + p.T[9] * syp + p.T[10] * szp + p.T[11];
    // d(cam_tgt) / d(depth) = R v
    const float rx = p.T[0] * vx + p.T[1] * vy + p.T[2] * vz, ry = p.T[4] * vx + p.T[5] * vy + p.T[6] * vz,
                rz = p.T[8] * vx + p.T[9] * vy + p.T[10] * vz;
    const float hx = p.K[0] * cx + p.K[1] * cy + p.K[2] * cz, hy = p.K[3] * cx + p.K[4] * cy + p.K[5] * cz,
                hz = p.K[6] * cx + p.K[7] * cy + p.K[8] * cz;
    const float gx = p.K[0] * rx + p.K[1] * ry + p.K[2] * rz, gy = p.K[3] * rx + p.K[4] * ry + p.K[5] * rz,
                gz = p.K[6] * rx + p.K[7] * ry + p.K[8] * rz;
    val = cz > 0.f ? 1.f : 0.f;
    const bool front = hz > 0.f;
    const float qx = front ? hx / hz : -1.f, qy = front ? hy / hz : -1.f;
    const float dqx = front ? (gx * hz - hx * gz) / (hz * hz) : 0.f, dqy = front ? (gy * hz - hy * gz) / (hz * hz) : 0.f;
    bilin3(p.img_t, p.H, p.W, qx, qy, ct, tx, ty);
    float l_rep = 0.f, l_id = 0.f, dl = 0.f;
    col_l1 = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e = ct[c] - cs[c];
        l_rep += fabsf(e);
        dl += (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) * (tx[c] * dqx + ty[c] * dqy);
        l_id += fabsf(ci[c] - cs[c]);
        const float col = p.color[3 * r + c];
        const float lc = fabsf(col - cs[c]);
        col_l1 += lc;
        if (p.loss_color) p.loss_color[3 * r + c] = lc;
        p.col_src[3 * r + c] = cs[c];
    }
    l_rep *= (1.f / 3.f);
    l_id = l_id * (1.f / 3.f) + (p.noise ? p.noise[r] * p.noise_scale : (p.rng ? sl_normal(p.rng, r) * p.noise_scale : 0.f));
    const bool rep = l_rep <= l_id;   // torch.minimum: the gradient goes to the reprojection term where it is the smaller (or equal)
    term = rep ? l_rep : l_id;
    if (p.ray_term) p.ray_term[r] = term;
    p.valid[r] = val;
    p.dterm_ddepth[r] = rep ? dl * (1.f / 3.f) : 0.f;
}

__global__ __launch_bounds__(256) void loss_side_fwd_kernel(LossArgs p) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    float term = 0.f, val = 0.f, cl = 0.f;
    if (r < p.R) loss_ray(p, r, term, val, cl);
    // masked mean: numerator / denominator per wave, then two atomics per wave; the last block to arrive divides
    float num = wave_sum(term * val), den = wave_sum(val);
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(p.acc, num);
        atomicAdd(p.acc + 1, den);
    }
}
__global__ void loss_side_mean_kernel(const float* acc, float* out) { out[0] = acc[0] / fmaxf(acc[1], 1.f); }

// backward: g_color = g_loss_color * sign(color - col_src); g_depth = g_loss_rep * valid / max(n_valid, 1) * dterm/ddepth
__global__ __launch_bounds__(256) void loss_side_bwd_kernel(const float* color, const float* col_src, const float* valid, const float* dterm,
                                                            const float* acc, const float* g_lc, const float* g_lr, int R, float* g_color,
                                                            float* g_depth) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float gl = g_lr ? g_lr[0] / fmaxf(acc[1], 1.f) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e = color[3 * r + c] - col_src[3 * r + c];
        g_color[3 * r + c] = g_lc ? g_lc[3 * r + c] * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 0.f;
    }
    g_depth[r] = gl * valid[r] * dterm[r];
}


// ---- the whole loss of one source frame (reference scenerf.py:203-238 around process_single_source :243-320) -------------------------
//   total = w_rep * loss_reprojection + w_col * mean(loss_color) + mean(loss_kl) + w_d2c * mean_r min_k |gaussian_means[r][k] - depth[r]|
// (depth detached in the last term, scenerf.py:287-290), plus the two means the trainer only logs (som_vars / gaussian_stds of the closest
// gaussian).  ~12 eager kernels forward and as many backward between the renderer's forward and backward -- on the critical path of a
// step, 6-7 us each -- become one launch each way (a second, one-wave launch sums the per-block partials when R > 1024).  Sums are
// taken per block in a fixed order (no atomics): the value is reproducible run to run.
#define SL_TERMS 8      // rep numerator, rep denominator, colour sum, kl sum, d2c sum, som_vars sum, stds sum, (unused)
#define SL_THREADS 64   // one wave per block: a ray's 36 image taps are 36 cache misses (two 5 MB images, random pixels), and what bounds the
                        // kernel is how many misses a CU keeps in flight -- 1,200 rays in two 1,024-thread blocks (two CUs) took 71 us, in
                        // nineteen one-wave blocks they spread over nineteen CUs (r04_f)

struct SrcLossArgs {
    LossArgs L;
    const float* loss_kl;    // [R]
    const float* gmeans;     // [R][G]
    const float* gstds;      // [R][G] or NULL
    const float* som_vars;   // [R][G] or NULL
    int G;
    float w_rep, w_col, w_d2c;
    int* closest;            // [R] index of the gaussian closest to the rendered depth (kept for the backward)
    float* partial;          // [blocks][SL_TERMS]
    float* out;              // [8]: total, loss_reprojection, mean loss_color, mean loss_kl, mean dist2closest, mean min_som_vars, mean min_stds, n_valid
    float* total;            // [1]: the total once more, in a buffer of its own (the differentiable output)
};

__device__ static inline void sl_finish(const SrcLossArgs& p, const float (&t)[SL_TERMS]) {
    const float invR = 1.f / (float)p.L.R;
    const float rep = t[0] / fmaxf(t[1], 1.f), col = t[2] * invR * (1.f / 3.f), kl = t[3] * invR, d2c = t[4] * invR;
    p.out[0] = p.w_rep * rep + p.w_col * col + kl + p.w_d2c * d2c;
    p.total[0] = p.out[0];
    p.out[1] = rep; p.out[2] = col; p.out[3] = kl; p.out[4] = d2c;
    p.out[5] = t[5] * invR; p.out[6] = t[6] * invR; p.out[7] = t[1];
    if (p.L.rng && !p.L.noise) ((unsigned long long*)p.L.rng)[1] += 1ull;     // the next call draws fresh noise (every block has read the counter)
}

__global__ __launch_bounds__(SL_THREADS) void source_loss_fwd_kernel(SrcLossArgs p) {
    __shared__ float s_part[SL_THREADS / 64][SL_TERMS];
    const int r = blockIdx.x * SL_THREADS + threadIdx.x;
    float t[SL_TERMS] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < p.L.R) {
        float term, val, cl;
        loss_ray(p.L, r, term, val, cl);
        t[0] = term * val; t[1] = val; t[2] = cl; t[3] = p.loss_kl[r];
        const float d = p.L.depth[r];
        float best = fabsf(p.gmeans[(size_t)r * p.G] - d);
        int bi = 0;
        for (int k = 1; k < p.G; ++k) {     // torch.min(dim=1): the first minimum
            const float v = fabsf(p.gmeans[(size_t)r * p.G + k] - d);
            if (v < best) { best = v; bi = k; }
        }
        p.closest[r] = bi;
        t[4] = best;
        t[5] = p.som_vars ? p.som_vars[(size_t)r * p.G + bi] : 0.f;
        t[6] = p.gstds ? p.gstds[(size_t)r * p.G + bi] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < SL_TERMS; ++i) t[i] = wave_sum(t[i]);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < SL_TERMS; ++i) s_part[wv][i] = t[i];
    }
    __syncthreads();
    if (threadIdx.x < SL_TERMS) {
        float a = 0.f;
        for (int w = 0; w < SL_THREADS / 64; ++w) a += s_part[w][threadIdx.x];
        if (gridDim.x > 1) p.partial[blockIdx.x * SL_TERMS + threadIdx.x] = a;
        s_part[0][threadIdx.x] = a;
    }
    if (gridDim.x == 1) {
        __syncthreads();
        if (threadIdx.x == 0) {
            float tt[SL_TERMS];
#pragma unroll
            for (int i = 0; i < SL_TERMS; ++i) tt[i] = s_part[0][i];
            sl_finish(p, tt);
        }
    }
}
__global__ void source_loss_finish_kernel(SrcLossArgs p, int blocks) {
    float tt[SL_TERMS];
#pragma unroll
    for (int i = 0; i < SL_TERMS; ++i) {
        float a = 0.f;
        for (int b = (int)threadIdx.x; b < blocks; b += 64) a += p.partial[b * SL_TERMS + i];
        tt[i] = wave_sum(a);
    }
    if (threadIdx.x == 0) sl_finish(p, tt);
}

// backward of the total w.r.t. colour, depth, loss_kl and the gaussian means; g = upstream gradient of the total (device scalar or NULL = 1)
__global__ __launch_bounds__(256) void source_loss_bwd_kernel(const float* color, const float* col_src, const float* valid, const float* dterm,
                                                              const float* gmeans, const float* depth, const int* closest, const float* out,
                                                              const float* g, int R, int G, float w_rep, float w_col, float w_d2c,
                                                              float* g_color, float* g_depth, float* g_kl, float* g_gmeans) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= R) return;
    const float gt = g ? g[0] : 1.f, invR = 1.f / (float)R;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float e = color[3 * r + c] - col_src[3 * r + c];
        g_color[3 * r + c] = gt * w_col * invR * (1.f / 3.f) * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f));
    }
    g_depth[r] = gt * w_rep / fmaxf(out[7], 1.f) * valid[r] * dterm[r];
    g_kl[r] = gt * invR;
    const int bi = closest[r];
    for (int k = 0; k < G; ++k) {
        const float e = gmeans[(size_t)r * G + k] - depth[r];
        g_gmeans[(size_t)r * G + k] = k == bi ? gt * w_d2c * invR * (e > 0.f ? 1.f : (e < 0.f ? -1.f : 0.f)) : 0.f;
    }
}

// ---- depth metrics of one evaluation (loss/depth_metrics.py:3-24): the seven means in one launch of one block ------------------------
#define DM_THREADS 1024
__global__ __launch_bounds__(DM_THREADS) void depth_errors_kernel(const float* __restrict__ gt, const float* __restrict__ pred,
                                                                  const unsigned char* __restrict__ mask, long long n, float min_depth,
                                                                  float max_depth, float* __restrict__ out) {
    __shared__ double s_part[DM_THREADS / 64][8];
    // {abs_rel, sq_rel, (gt - pred)^2, (log gt - log pred)^2, [thresh < 1.25], [< 1.25^2], [< 1.25^3], count}: per-thread sums in double
    double t[8] = {0., 0., 0., 0., 0., 0., 0., 0.};
    const float th1 = 1.25f, th2 = 1.25f * 1.25f, th3 = 1.25f * 1.25f * 1.25f;
    for (long long i = threadIdx.x; i < n; i += DM_THREADS) {
        if (mask && !mask[i]) continue;
        const float g = gt[i];
        const float p = fminf(fmaxf(pred[i], min_depth), max_depth);
        const float d = g - p, thr = fmaxf(g / p, p / g), dl = logf(g) - logf(p);
        t[0] += (double)(fabsf(d) / g); t[1] += (double)(d * d / g); t[2] += (double)(d * d); t[3] += (double)(dl * dl);
        t[4] += thr < th1 ? 1. : 0.; t[5] += thr < th2 ? 1. : 0.; t[6] += thr < th3 ? 1. : 0.; t[7] += 1.;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        double v = t[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        t[k] = v;
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) s_part[wv][k] = t[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a[8];
        for (int k = 0; k < 8; ++k) {
            a[k] = 0.;
            for (int w = 0; w < DM_THREADS / 64; ++w) a[k] += s_part[w][k];
        }
        const double den = a[7] > 1. ? a[7] : 1.;      // (an empty mask: all zeros, where the reference skips the logging)
        out[0] = (float)(a[0] / den); out[1] = (float)(a[1] / den);
        out[2] = (float)sqrt(a[2] / den); out[3] = (float)sqrt(a[3] / den);
        out[4] = (float)(a[4] / den); out[5] = (float)(a[5] / den); out[6] = (float)(a[6] / den);
        out[7] = (float)a[7];
    }
}

extern "C" {

int scenerf_hip_loss_side_forward(const float* pix, const float* color, const float* depth, const float* img_source,
                                  const float* img_target, const float* noise, const float* cam_K, const float* inv_K,
                                  const float* T_source2target, int R, int H, int W, float* loss_color, float* ray_term, float* valid,
                                  float* dterm_ddepth, float* col_src, float* acc2, float* loss_reprojection, scenerf_stream_t stream) {
    SRF_CHECK(pix && color && depth && img_source && img_target && cam_K && inv_K && T_source2target && loss_color && ray_term && valid &&
                  dterm_ddepth && col_src && acc2 && loss_reprojection && R > 0 && H > 1 && W > 1,
              "loss_side_forward: NULL / empty argument");
    LossArgs p = {};
    p.pix = pix; p.color = color; p.depth = depth; p.img_s = img_source; p.img_t = img_target; p.noise = noise; p.noise_scale = 1.f;
    p.K = cam_K; p.invK = inv_K; p.T = T_source2target;   // device memory (wave-uniform scalar loads): no host round trip per call
    p.R = R; p.H = H; p.W = W;
    p.loss_color = loss_color; p.ray_term = ray_term; p.valid = valid; p.dterm_ddepth = dterm_ddepth; p.col_src = col_src;
    p.loss_rep = loss_reprojection; p.acc = acc2;
    hipStream_t s = as_stream(stream);
    SRF_HIP(hipMemsetAsync(acc2, 0, 2 * sizeof(float), s));
    SrfLaunchScope ps(s, "loss_side_fwd", 0, 0);
    loss_side_fwd_kernel<<<cdiv(R, 256), 256, 0, s>>>(p);
    loss_side_mean_kernel<<<1, 1, 0, s>>>(acc2, loss_reprojection);
    SRF_LAUNCH_CHECK("loss_side_fwd_kernel");
    return 0;
}

int scenerf_hip_loss_side_backward(const float* color, const float* col_src, const float* valid, const float* dterm_ddepth,
                                   const float* acc2, const float* g_loss_color, const float* g_loss_reprojection, int R,
                                   float* g_color, float* g_depth, scenerf_stream_t stream) {
    SRF_CHECK(color && col_src && valid && dterm_ddepth && acc2 && g_color && g_depth && R > 0, "loss_side_backward: NULL / empty argument");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "loss_side_bwd", 0, 0);
    loss_side_bwd_kernel<<<cdiv(R, 256), 256, 0, s>>>(color, col_src, valid, dterm_ddepth, acc2, g_loss_color, g_loss_reprojection, R, g_color,
                                                       g_depth);
    SRF_LAUNCH_CHECK("loss_side_bwd_kernel");
    return 0;
}


int scenerf_hip_source_loss_forward(const float* pix, const float* color, const float* depth, const float* loss_kl, const float* gmeans,
                                    const float* gstds, const float* som_vars, int G, const float* img_source, const float* img_target,
                                    const float* noise, uint64_t* rng_state, float noise_scale, const float* cam_K, const float* inv_K,
                                    const float* T_source2target, int R, int H, int W, float w_rep, float w_col, float w_d2c, float* valid,
                                    float* dterm_ddepth, float* col_src, int32_t* closest, float* partial, float* out8,
                                    float* total, scenerf_stream_t stream) {
    SRF_CHECK(pix && color && depth && loss_kl && gmeans && img_source && img_target && cam_K && inv_K && T_source2target && valid &&
                  dterm_ddepth && col_src && closest && partial && out8 && total && R > 0 && H > 1 && W > 1 && G >= 1 && G <= SCENERF_MAX_GAUSSIANS,
              "source_loss_forward: NULL / empty argument");
    SrcLossArgs p = {};
    p.L.pix = pix; p.L.color = color; p.L.depth = depth; p.L.img_s = img_source; p.L.img_t = img_target; p.L.noise = noise; p.L.noise_scale = noise_scale;
    p.L.rng = (const unsigned long long*)rng_state;
    p.L.K = cam_K; p.L.invK = inv_K; p.L.T = T_source2target;
    p.L.R = R; p.L.H = H; p.L.W = W;
    p.L.valid = valid; p.L.dterm_ddepth = dterm_ddepth; p.L.col_src = col_src;    // (loss_color / ray_term: not materialised)
    p.loss_kl = loss_kl; p.gmeans = gmeans; p.gstds = gstds; p.som_vars = som_vars; p.G = G;
    p.w_rep = w_rep; p.w_col = w_col; p.w_d2c = w_d2c;
    p.closest = closest; p.partial = partial; p.out = out8; p.total = total;
    hipStream_t s = as_stream(stream);
    const int blocks = cdiv(R, SL_THREADS);
    SrfLaunchScope ps(s, "source_loss_fwd", 0, 0);
    source_loss_fwd_kernel<<<blocks, SL_THREADS, 0, s>>>(p);
    if (blocks > 1) source_loss_finish_kernel<<<1, 64, 0, s>>>(p, blocks);
    SRF_LAUNCH_CHECK("source_loss_fwd_kernel");
    return 0;
}

int scenerf_hip_source_loss_backward(const float* color, const float* col_src, const float* valid, const float* dterm_ddepth,
                                     const float* gmeans, const float* depth, const int32_t* closest, const float* out8, const float* g_total,
                                     int R, int G, float w_rep, float w_col, float w_d2c, float* g_color, float* g_depth, float* g_loss_kl,
                                     float* g_gmeans, scenerf_stream_t stream) {
    SRF_CHECK(color && col_src && valid && dterm_ddepth && gmeans && depth && closest && out8 && g_color && g_depth && g_loss_kl && g_gmeans &&
                  R > 0 && G >= 1, "source_loss_backward: NULL / empty argument");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "source_loss_bwd", 0, 0);
    source_loss_bwd_kernel<<<cdiv(R, 256), 256, 0, s>>>(color, col_src, valid, dterm_ddepth, gmeans, depth, closest, out8, g_total, R, G, w_rep,
                                                         w_col, w_d2c, g_color, g_depth, g_loss_kl, g_gmeans);
    SRF_LAUNCH_CHECK("source_loss_bwd_kernel");
    return 0;
}

int scenerf_hip_depth_errors(const float* gt, const float* pred, const unsigned char* mask, int64_t n, float min_depth, float max_depth,
                             float* out8, scenerf_stream_t stream) {
    SRF_CHECK(gt && pred && out8 && n >= 0, "depth_errors: bad args");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "depth_errors", 0, (double)n * 9);
    depth_errors_kernel<<<1, DM_THREADS, 0, s>>>(gt, pred, mask, (long long)n, min_depth, max_depth, out8);
    SRF_LAUNCH_CHECK("depth_errors_kernel");
    return 0;
}

}  // extern "C"
