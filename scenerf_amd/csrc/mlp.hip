// ResnetFC forward / backward as a sequence of MFMA GEMM launches with fused prologues/epilogues
// (reference scenerf/models/resnetfc.py:133-164 and its autograd), plus the small memory-bound kernels
// around them: lin_out (512 -> 4|2), bias-gradient column sums and an fp32->bf16 row conversion.
#include <stdlib.h>

#include "fused.h"


// ------------------------------------------------------------------------------------------------ lin_out
// logits[m][j] = relu(H3[m][:]) . w_out[j][:] + b_out[j]; one wave per row, lane owns 8 of the 512 columns
template <typename T, int DO>
__global__ __launch_bounds__(256) void linout_fwd_kernel(const void* __restrict__ H3, const float* __restrict__ w_out,
                                                         const float* __restrict__ b_out, int M, float* __restrict__ logits) {
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    float w[DO][8];
#pragma unroll
    for (int j = 0; j < DO; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[j][e] = w_out[j * SCENERF_D_HIDDEN + lane * 8 + e];
    for (int m = wave; m < M; m += nwaves) {
        float h[8];
        load8<T>(H3, (size_t)m * SCENERF_D_HIDDEN + lane * 8, h);
        float acc[DO];
#pragma unroll
        for (int j = 0; j < DO; ++j) {
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(fmaxf(h[e], 0.f), w[j][e], a);
            acc[j] = wave_sum(a);
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < DO; ++j) logits[(size_t)m * DO + j] = acc[j] + b_out[j];
        }
    }
}

// dH3[m][k] = [H3>0] * sum_j dl[m][j] w_out[j][k];  dw_out[j][k] += sum_m dl[m][j] relu(H3[m][k]);  db_out[j] += sum_m dl[m][j]
// one wave per row, lane owns 8 of the 512 columns; 8 rows per iteration so 8 independent 16-byte loads are in flight (r06: 39.5 us against 42 with 4, 46.8 with 2, for lin_out's weight gradient at 153,600 rows)
// DH = false: the weight / bias gradients only (the 128-row dgrad chain makes dH3 itself: wide.hip MODE 1), H3 is read and nothing but
// the partial sums is written
template <typename T, int DO, bool DH>
__global__ __launch_bounds__(256) void linout_bwd_kernel(const void* __restrict__ H3, const float* __restrict__ w_out,
                                                         const float* __restrict__ dlog, int M, void* __restrict__ dH3, int lddh,
                                                         float* __restrict__ partial) {
    __shared__ float s_dw[4][DO][SCENERF_D_HIDDEN];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wave = blockIdx.x * 4 + wv, nwaves = gridDim.x * 4;
    float w[DO][8], dw[DO][8], db[DO];
#pragma unroll
    for (int j = 0; j < DO; ++j) {
        db[j] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            w[j][e] = w_out[j * SCENERF_D_HIDDEN + lane * 8 + e];
            dw[j][e] = 0.f;
        }
    }
#ifndef LOB_RU
#define LOB_RU 8
#endif
    constexpr int RU = LOB_RU;
    for (int mb = wave * RU; mb < M; mb += nwaves * RU) {
        float h[RU][8], dl[RU][DO];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int m = min(mb + u, M - 1);
            load8<T>(H3, (size_t)m * SCENERF_D_HIDDEN + lane * 8, h[u]);
#pragma unroll
            for (int j = 0; j < DO; ++j) dl[u][j] = (mb + u < M) ? dlog[(size_t)m * DO + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            float g[8];
#pragma unroll
            for (int j = 0; j < DO; ++j) db[j] += dl[u][j];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < DO; ++j) {
                    if (DH) a = fmaf(dl[u][j], w[j][e], a);
                    dw[j][e] = fmaf(dl[u][j], fmaxf(h[u][e], 0.f), dw[j][e]);
                }
                g[e] = h[u][e] > 0.f ? a : 0.f;
            }
            if (DH && mb + u < M) store8<T>(dH3, (size_t)(mb + u) * lddh + lane * 8, g);
        }
    }
#pragma unroll
    for (int j = 0; j < DO; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) s_dw[wv][j][lane * 8 + e] = dw[j][e];
    __syncthreads();
    // per-block partial sums go to scratch (no atomics: with ~1000 blocks adding into the same 2048 addresses the
    // same-address atomics serialised and dominated the kernel); linout_reduce_kernel folds them afterwards
    float* part = partial + (size_t)blockIdx.x * (DO * SCENERF_D_HIDDEN + 8);
    for (int i = threadIdx.x; i < DO * SCENERF_D_HIDDEN; i += 256) {
        int j = i / SCENERF_D_HIDDEN, k = i - j * SCENERF_D_HIDDEN;
        part[i] = s_dw[0][j][k] + s_dw[1][j][k] + s_dw[2][j][k] + s_dw[3][j][k];
    }
    __shared__ float s_db[4][DO];
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < DO; ++j) s_db[wv][j] = db[j];
    }
    __syncthreads();
    if (threadIdx.x < DO) part[DO * SCENERF_D_HIDDEN + threadIdx.x] = s_db[0][threadIdx.x] + s_db[1][threadIdx.x] + s_db[2][threadIdx.x] + s_db[3][threadIdx.x];
}

// dw_out[i] += sum over blocks of the partials.  grid.y slices of the block range run in parallel (a single thread per output
// walking ~1200 partials was latency-bound: 107 us), each adding its share with one atomic per output.
template <int DO>
__global__ void linout_reduce_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ dw_out, float* __restrict__ db_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int stride = DO * SCENERF_D_HIDDEN + 8;
    if (i >= DO * SCENERF_D_HIDDEN + DO) return;
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(nblocks, b0 + per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = b0;
    for (; b + 3 < b1; b += 4) {
        a0 += partial[(size_t)b * stride + i];
        a1 += partial[(size_t)(b + 1) * stride + i];
        a2 += partial[(size_t)(b + 2) * stride + i];
        a3 += partial[(size_t)(b + 3) * stride + i];
    }
    for (; b < b1; ++b) a0 += partial[(size_t)b * stride + i];
    const float v = (a0 + a1) + (a2 + a3);
    if (b1 > b0) unsafeAtomicAdd(i < DO * SCENERF_D_HIDDEN ? dw_out + i : db_out + (i - DO * SCENERF_D_HIDDEN), v);
}

// bf16 mode: lin_in runs inside the first hidden GEMM as three extra K-segments.  x = hi + lo with hi = bf16(x),
// lo = bf16(x - hi) (16 significant bits: raw xyz reaches ~100 m), and the weight is split the same way on the host;
// x_hi.w_hi + x_lo.w_hi + x_hi.w_lo reproduces the fp32 product to ~2^-16.  Row layout: [hi(48) | lo(48) | hi(48)].
__global__ void split_xenc_kernel(const float* __restrict__ in, bf16_t* __restrict__ out, int M) {
    // one thread per 8 consecutive encodings of a row: two 16-byte loads, three 16-byte stores
    static_assert(SCENERF_D_XENC % 8 == 0, "8 encodings per thread");
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)M * (SCENERF_D_XENC / 8)) return;
    const size_t m = i / (SCENERF_D_XENC / 8);
    const int c = (int)(i - m * (SCENERF_D_XENC / 8)) * 8;
    const float4 x0 = *(const float4*)(in + m * SCENERF_D_XENC + c), x1 = *(const float4*)(in + m * SCENERF_D_XENC + c + 4);
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf16_t h0 = f32_to_bf16(x[2 * e]), h1 = f32_to_bf16(x[2 * e + 1]);
        const bf16_t l0 = f32_to_bf16(x[2 * e] - bf16_to_f32(h0)), l1 = f32_to_bf16(x[2 * e + 1] - bf16_to_f32(h1));
        hi[e] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        lo[e] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    bf16_t* o = out + m * (3 * SCENERF_D_XENC) + c;
    *(uint4*)o = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    *(uint4*)(o + SCENERF_D_XENC) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    *(uint4*)(o + 2 * SCENERF_D_XENC) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
}

template <typename T>
static int launch_linout_fwd(int d_out, const void* H3, const float* w, const float* b, int M, float* logits, hipStream_t s) {
    int grid = cdiv(M, 4 * 16);
    if (grid > 2048) grid = 2048;
    SrfLaunchScope ps(s, "linout_fwd", 0, (double)M * (512.0 * sizeof(T) + 4.0 * d_out));
    if (d_out == 4) linout_fwd_kernel<T, 4><<<grid, 256, 0, s>>>(H3, w, b, M, logits);
    else linout_fwd_kernel<T, 2><<<grid, 256, 0, s>>>(H3, w, b, M, logits);
    SRF_LAUNCH_CHECK("linout_fwd_kernel");
    return 0;
}
// dH3 == NULL: weight / bias gradients only
template <typename T>
static int launch_linout_bwd(int d_out, const void* H3, const float* w, const float* dlog, int M, void* dH3, int lddh, float* dw,
                             float* db, float* scratch, hipStream_t s) {
    int grid = cdiv(M, 128);   // 32 rows per wave
    if (grid > 2048) grid = 2048;
    {
        SrfLaunchScope ps(s, dH3 ? "linout_bwd" : "linout_wgrad", 0, (double)M * ((dH3 ? 1024.0 : 512.0) * sizeof(T) + 4.0 * d_out));
        if (dH3) {
            if (d_out == 4) linout_bwd_kernel<T, 4, true><<<grid, 256, 0, s>>>(H3, w, dlog, M, dH3, lddh, scratch);
            else linout_bwd_kernel<T, 2, true><<<grid, 256, 0, s>>>(H3, w, dlog, M, dH3, lddh, scratch);
        } else {
            if (d_out == 4) linout_bwd_kernel<T, 4, false><<<grid, 256, 0, s>>>(H3, w, dlog, M, nullptr, 0, scratch);
            else linout_bwd_kernel<T, 2, false><<<grid, 256, 0, s>>>(H3, w, dlog, M, nullptr, 0, scratch);
        }
        SRF_LAUNCH_CHECK("linout_bwd_kernel");
    }
    const int n = d_out * SCENERF_D_HIDDEN + d_out;
    const dim3 rgrid(cdiv(n, 256), grid >= 64 ? 32 : 1);
    if (d_out == 4) linout_reduce_kernel<4><<<rgrid, 256, 0, s>>>(scratch, grid, dw, db);
    else linout_reduce_kernel<2><<<rgrid, 256, 0, s>>>(scratch, grid, dw, db);
    SRF_LAUNCH_CHECK("linout_reduce_kernel");
    return 0;
}

// ---- internal fork/join: weight-gradient GEMMs run on a side stream next to the dgrad chain -------------------
// The wgrad of a layer only needs that layer's incoming gradient, not the rest of the backward chain, so it can overlap
// the next dgrad GEMM: two kernels in flight hide each other's tails, epilogues and launch gaps.  The side stream is
// created once per caller stream; every call forks from and joins back into the caller's stream with events, so from
// the outside all work is still ordered on `stream` (and the pattern is hipGraph-capturable).
#include <map>
struct SideCtx {
    hipStream_t side = nullptr;
    hipEvent_t ev[16];
    int next = 0;
};
static SideCtx* side_ctx(hipStream_t main) {
    static std::map<hipStream_t, SideCtx*> g_map;   // keyed by the caller's stream (a stream belongs to one device)
    static std::mutex g_mu;
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_map.find(main);
    if (it != g_map.end()) return it->second;
    SideCtx* c = new SideCtx();
    if (hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess) { delete c; return nullptr; }
    for (int i = 0; i < 16; ++i) {
        if (hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) != hipSuccess) { delete c; return nullptr; }
    }
    g_map[main] = c;
    return c;
}
static int order_after(SideCtx* c, hipStream_t from, hipStream_t to) {  // everything enqueued later on `to` waits for `from`'s past
    hipEvent_t e = c->ev[c->next];
    c->next = (c->next + 1) & 15;
    SRF_HIP(hipEventRecord(e, from));
    SRF_HIP(hipStreamWaitEvent(to, e, 0));
    return 0;
}

// ================================================================================================ sequencing
static void set_segments(GemmNT& g, const scenerf_cfg* cfg, const void* Z, const uint8_t* tile_mask) {
    g.A2 = Z;
    g.lda2 = SCENERF_D_LATENT;
    g.nseg = 5;
    int off = 0;
    for (int s = 0; s < 5; ++s) {
        g.seg_off[s] = off;
        g.seg_len[s] = cfg->map_C[s];
        off += cfg->map_C[s];
    }
    g.tile_mask = tile_mask;
}

// dZ[:, slice_s] = dH[:, 0:1536] @ Wz[:, slice_s], scattered straight into the (H,W,C) map gradients (grid_sampler_2d_backward)
static int feature_grads(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const uint8_t* tile_mask, const int32_t* tap_texel,
                         const float* tap_weight, int M, const void* dH, float* const gmaps_hwc[SCENERF_N_SCALES], hipStream_t s) {
    const bool head = w->d_out == 2;
    const bool per_scale = (cfg->flags & SCENERF_FLAG_DFEAT_PER_SCALE) != 0;   // one launch per scale (the older form), for A/B runs
    // bf16: the dedicated kernel (dfeat.hip: dH read once, tile-level texel reduction in LDS); fp32 and the A/B flags: the GEMM family's
    // scatter epilogue
    // (one workgroup per 128-row tile: needs enough tiles to fill the chip, and rows whose taps run along a ray -- the gaussian
    // head's 4,800 anchor rows are 38 tiles of unrelated texels: 270 us there against 82 us through the GEMM)
    if (cfg->precision && !per_scale && !(cfg->flags & SCENERF_FLAG_DFEAT_GEMM) && cdiv(M, 128) >= SRF_WIDE_MIN_BLOCKS)
        return launch_dfeat_scatter(cfg, w, tile_mask, tap_texel, tap_weight, M, dH, gmaps_hwc, s);
    GemmNT g;
    g.name = head ? "gemm_dfeat_scatter/g" : "gemm_dfeat_scatter";
    g.A1 = dH; g.lda1 = 4 * SCENERF_D_HIDDEN; g.K1 = 3 * SCENERF_D_HIDDEN;
    g.ldw = 3 * SCENERF_D_HIDDEN;
    g.M = M;
    g.tile_mask = tile_mask;
    g.tap_texel = tap_texel; g.tap_weight = tap_weight;
    if (!per_scale) {
        // ONE launch for all scales: the row tile's dH (K = 1536: 393 KB per 128 rows) is read from HBM once and feeds the column
        // tiles of every scale it touches back to back on one XCD (five launches re-read all of dH, 472 MB, each)
        bool any = false;
        int t = 0;
        for (int sc = 0; sc < 5; ++sc) {
            g.ms_t0[sc] = t;
            g.ms_C[sc] = cfg->map_C[sc];
            g.ms_W[sc] = w->w_z_t[sc];
            g.ms_gmap[sc] = gmaps_hwc[sc];
            if (cfg->map_chw[sc] == 1) { g.ms_st[sc] = 1; g.ms_sc[sc] = (long)cfg->map_H[sc] * cfg->map_W[sc]; }   // (C,H,W) gradient buffer
            any = any || gmaps_hwc[sc];
            t += cdiv(cfg->map_C[sc], 128);
        }
        if (!any) return 0;
        g.ms_t0[5] = t;
        g.ms_n = 5;
        g.N = SCENERF_D_LATENT;
        g.scatter_scale = 0;
        return launch_gemm_nt(cfg->precision, g, s);
    }
    for (int sc = 0; sc < 5; ++sc) {
        if (!gmaps_hwc[sc]) continue;
        g.W = w->w_z_t[sc];
        g.N = cfg->map_C[sc];
        g.skip_bit = sc;
        g.gmap = gmaps_hwc[sc]; g.scatter_scale = sc;
        g.gmap_st = 0; g.gmap_sc = 1;
        if (cfg->map_chw[sc] == 1) { g.gmap_st = 1; g.gmap_sc = (long)cfg->map_H[sc] * cfg->map_W[sc]; }   // (C,H,W) gradient buffer
        if (int e = launch_gemm_nt(cfg->precision, g, s)) return e;
    }
    return 0;
}

// one wave that sleeps ~`us` microseconds (s_sleep 31 = 1,984 cycles, the clock holds ~2 GHz under load): a stream-ordered head start for
// a launch on another stream
__global__ void srf_delay_kernel(int us) {
    for (int i = 0; i < us; ++i) __builtin_amdgcn_s_sleep(31);
}

// blocks 0..5: dst[0:512] = a, dst[512:1024] = b, dst[1024:1536] = c ; blocks 6..: w_dense[512][42] = w_in[512][SCENERF_WIN_LD][:, 0:42]
// (lin_in.weight's gradient as the dense tensor autograd wants: handed over as a column slice of the 256-wide sink, AccumulateGrad
// cloned it -- one more launch per MLP on the step's critical path, in front of the optimizer)
__global__ void copy3_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                             float* __restrict__ w_dense, const float* __restrict__ w_in) {
    if (blockIdx.x >= 3 * SCENERF_D_HIDDEN / 256) {
        const int i = (blockIdx.x - 3 * SCENERF_D_HIDDEN / 256) * 256 + threadIdx.x;
        if (w_dense && i < SCENERF_D_HIDDEN * 42) w_dense[i] = w_in[(i / 42) * SCENERF_WIN_LD + i % 42];
        return;
    }
    const int i = blockIdx.x * 256 + threadIdx.x, j = i & (SCENERF_D_HIDDEN - 1);
    dst[i] = i < SCENERF_D_HIDDEN ? a[j] : (i < 2 * SCENERF_D_HIDDEN ? b[j] : c[j]);
}

extern "C" {

int scenerf_hip_prepare(const scenerf_cfg* cfg, scenerf_stream_t stream) {
    SRF_CHECK(cfg, "prepare: cfg is NULL");
    hipStream_t s = as_stream(stream);
    if (int e = gemm_prepare()) return e;
    if (int e = wgrad_prepare()) return e;
    if (cfg->precision) {
        if (int e = fused_prepare(cfg, s)) return e;
        if (int e = wide_prepare(cfg, s)) return e;
    }
    return 0;
}

int scenerf_hip_mlp_feature_grads(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const uint8_t* tile_mask,
                                  const int32_t* tap_texel, const float* tap_weight, int M, const void* dH,
                                  float* const gmaps_hwc[SCENERF_N_SCALES], scenerf_stream_t stream) {
    SRF_CHECK(cfg && w && tile_mask && tap_texel && tap_weight && dH && gmaps_hwc && M > 0, "mlp_feature_grads: NULL argument");
    // SCENERF_FLAG_WGRAD_OVERLAP here = "the caller runs this beside the batched weight gradients of the same pass, on another stream":
    // the head start that launch needs (see scenerf_hip_mlp_backward) is given on this stream
    if ((cfg->flags & SCENERF_FLAG_WGRAD_OVERLAP) && srf_dfeat_delay_us() > 0 && w->d_out != 2 && cfg->precision && cdiv(M, 128) >= SRF_WIDE_MIN_BLOCKS) {
        srf_delay_kernel<<<1, 64, 0, as_stream(stream)>>>(srf_dfeat_delay_us());
        SRF_LAUNCH_CHECK("srf_delay_kernel");
    }
    return feature_grads(cfg, w, tile_mask, tap_texel, tap_weight, M, dH, gmaps_hwc, as_stream(stream));
}

int scenerf_hip_mlp_forward(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const float* xenc,
                            const uint8_t* tile_mask, int M, const scenerf_mlp_acts* a, scenerf_stream_t stream) {
    SRF_CHECK(cfg && w && Z && tile_mask && a && M > 0 && (xenc || (cfg->precision == 1 && a->x3_ready)), "mlp_forward: NULL argument");
    SRF_CHECK(w->d_out == 4 || w->d_out == 2, "mlp_forward: d_out must be 4 or 2");
    const int prec = cfg->precision;
    const bool head = w->d_out == 2;  // profile names: ".../g" = gaussian head (4 points per ray)
    hipStream_t s = as_stream(stream);
    if (prec == 0) {
        // fp32 mode: lin_in as its own fp32-MFMA GEMM, then H0 = h0pre + lin_z.0(z)
        {
            GemmNT g;
            g.name = head ? "gemm_lin_in/g" : "gemm_lin_in";
            g.A1 = xenc; g.lda1 = SCENERF_D_XENC; g.K1 = SCENERF_D_XENC;
            g.W = w->w_in; g.ldw = SCENERF_D_XENC;
            g.M = M; g.N = SCENERF_D_HIDDEN; g.bias = w->b_in;
            g.out = a->h0pre; g.ldout = SCENERF_D_HIDDEN; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
        GemmNT g;
        g.name = head ? "gemm_fwd_linz0/g" : "gemm_fwd_linz0";
        set_segments(g, cfg, Z, tile_mask);
        g.W = w->w_h[0]; g.ldw = SCENERF_D_LATENT;
        g.M = M; g.N = SCENERF_D_HIDDEN; g.bias = w->b_h[0];
        g.res = a->h0pre; g.ldres = SCENERF_D_HIDDEN; g.res_f32 = 1;
        g.out = a->H[0]; g.ldout = SCENERF_D_HIDDEN;
        if (int e = launch_gemm_nt(prec, g, s)) return e;
    } else {
        // bf16 mode: H0 = [x_hi | x_lo | x_hi | z] @ [w_hi | w_hi | w_lo | lin_z.0]^T + (lin_in.bias + lin_z.0.bias); the
        // split encoding lives in the (otherwise unused) h0pre scratch and is reused by the lin_in weight gradient
        bf16_t* x3 = (bf16_t*)a->h0pre;
        if (!a->x3_ready) {   // (scenerf_hip_encode_points writes the split encoding itself when it is handed the buffer)
            SRF_CHECK(xenc, "mlp_forward: xenc is NULL and acts->x3_ready is not set");
            SrfLaunchScope ps(s, "split_xenc", 0, (double)M * SCENERF_D_XENC * 10);
            size_t n = (size_t)M * (SCENERF_D_XENC / 8);
            split_xenc_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(xenc, x3, M);
            SRF_LAUNCH_CHECK("split_xenc_kernel");
        }
        // rows from which the whole trunk runs as ONE fused kernel (scenerf_cfg.fused_min_rows), lin_out included
        if (srf_use_fused(cfg, M) && w->w_stream) {
            // scenerf_cfg.fwd_kernel: 0 = fused.hip's 64-row LDS-ring pipeline, 2 = wide.hip's 128-row blocks (1 was a register-streamed
            // 64-row variant with identical results and the same speed: removed in round 3)
            SRF_CHECK(cfg->fwd_kernel == 0 || cfg->fwd_kernel == 2, "mlp_forward: unknown fwd_kernel %d", cfg->fwd_kernel);
            // (128-row blocks need enough of them to fill the chip: the gaussian head's 4,800 rows are 38 blocks on 256 CUs -- 156 us against
            // the 64-row ring kernel's 109 us)
            if (cfg->fwd_kernel == 2 && (cdiv(M, 128) >= SRF_WIDE_MIN_BLOCKS || (cfg->flags & SCENERF_FLAG_WIDE_ANY_M))) return launch_mlp_fwd_wide(cfg, w, Z, tile_mask, M, a, s);
            return launch_mlp_fwd_fused(cfg, w, Z, tile_mask, M, a, s);
        }
        SRF_CHECK(a->H[3], "mlp_forward: acts->H[3] is NULL");
        SRF_CHECK(a->H[0] && a->H[1] && a->H[2] && a->Nn[0] && a->Nn[1] && a->Nn[2],
                  "mlp_forward: the per-layer path needs every activation buffer (NULL entries are for fused-kernel inference only)");
        GemmNT g;
        g.name = head ? "gemm_fwd_in_linz0/g" : "gemm_fwd_in_linz0";
        g.A1 = x3; g.lda1 = 3 * SCENERF_D_XENC; g.K1 = 3 * SCENERF_D_XENC;
        set_segments(g, cfg, Z, tile_mask);
        g.W = w->w_h[0]; g.ldw = 3 * SCENERF_D_XENC + SCENERF_D_LATENT;
        g.M = M; g.N = SCENERF_D_HIDDEN; g.bias = w->b_h[0];
        g.out = a->H[0]; g.ldout = SCENERF_D_HIDDEN;
        if (int e = launch_gemm_nt(prec, g, s)) return e;
    }
    for (int b = 0; b < 3; ++b) {
        {   // net = fc_0(relu(h))
            GemmNT g;
            g.name = head ? "gemm_fwd_fc0/g" : "gemm_fwd_fc0";
            g.A1 = a->H[b]; g.lda1 = SCENERF_D_HIDDEN; g.K1 = SCENERF_D_HIDDEN; g.relu1 = 1;
            g.W = w->w_fc0[b]; g.ldw = SCENERF_D_HIDDEN;
            g.M = M; g.N = SCENERF_D_HIDDEN; g.bias = w->b_fc0[b];
            g.out = a->Nn[b]; g.ldout = SCENERF_D_HIDDEN;
            if (int e = launch_gemm_nt(prec, g, s)) return e;
        }
        {   // h = h + fc_1(relu(net)) [+ lin_z.(b+1)(z)]
            GemmNT g;
            g.name = b < 2 ? (head ? "gemm_fwd_fc1_linz/g" : "gemm_fwd_fc1_linz") : (head ? "gemm_fwd_fc1/g" : "gemm_fwd_fc1");
            g.A1 = a->Nn[b]; g.lda1 = SCENERF_D_HIDDEN; g.K1 = SCENERF_D_HIDDEN; g.relu1 = 1;
            g.ldw = SCENERF_D_HIDDEN;
            if (b < 2) {
                set_segments(g, cfg, Z, tile_mask);
                g.ldw = SCENERF_D_HIDDEN + SCENERF_D_LATENT;
            }
            g.W = w->w_h[b + 1];
            g.M = M; g.N = SCENERF_D_HIDDEN; g.bias = w->b_h[b + 1];
            g.res = a->H[b]; g.ldres = SCENERF_D_HIDDEN;
            g.out = a->H[b + 1]; g.ldout = SCENERF_D_HIDDEN;
            if (int e = launch_gemm_nt(prec, g, s)) return e;
        }
    }
    if (prec) return launch_linout_fwd<bf16_t>(w->d_out, a->H[3], w->w_out, w->b_out, M, a->logits, s);
    return launch_linout_fwd<float>(w->d_out, a->H[3], w->w_out, w->b_out, M, a->logits, s);
}

// ResnetFC.forward (resnetfc.py:133-164) for ANY block count / hidden width, fp32, forward only: one fp32-MFMA GEMM per nn.Linear, the
// reference's order of additions (h = lin_in(x); per block: h += lin_z.b(z); h += fc_1(relu(fc_0(relu(h)))); out = lin_out(relu(h))).
// The product's fast kernels are specialised to the trunk SceneRF instantiates (3 blocks x 512); this entry is what makes every other
// ResnetFC shape -- BASELINE.json configs[0]'s 1 block x 128 -- run through the same ray pipeline on the GPU.
int scenerf_hip_resnetfc_forward(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const float* xenc, const float* Z,
                                 const uint8_t* tile_mask, int M, float* h_a, float* h_b, float* n_buf, float* logits,
                                 scenerf_stream_t stream) {
    SRF_CHECK(cfg && net && xenc && Z && tile_mask && h_a && h_b && n_buf && logits && M > 0, "resnetfc_forward: NULL argument");
    SRF_CHECK(net->n_blocks >= 1 && net->n_blocks <= SCENERF_RESNETFC_MAX_BLOCKS, "resnetfc_forward: n_blocks=%d (1..%d)", net->n_blocks,
              SCENERF_RESNETFC_MAX_BLOCKS);
    SRF_CHECK(net->d_hidden >= 16 && net->d_hidden % 16 == 0 && net->d_out_pad >= 8 && net->d_out_pad % 8 == 0,
              "resnetfc_forward: d_hidden=%d must be a multiple of 16, d_out_pad=%d of 8", net->d_hidden, net->d_out_pad);
    SRF_CHECK(net->w_in && net->b_in && net->w_out && net->b_out, "resnetfc_forward: NULL parameter");
    hipStream_t s = as_stream(stream);
    const int H = net->d_hidden;
    float* cur = h_a;
    float* nxt = h_b;
    {
        GemmNT g;
        g.name = "gemm_generic_lin_in";
        g.A1 = xenc; g.lda1 = SCENERF_D_XENC; g.K1 = SCENERF_D_XENC;
        g.W = net->w_in; g.ldw = SCENERF_D_XENC;
        g.M = M; g.N = H; g.bias = net->b_in;
        g.out = cur; g.ldout = H; g.out_f32 = 1;
        if (int e = launch_gemm_nt(0, g, s)) return e;
    }
    for (int b = 0; b < net->n_blocks; ++b) {
        SRF_CHECK(net->w_z[b] && net->b_z[b] && net->w_fc0[b] && net->b_fc0[b] && net->w_fc1[b] && net->b_fc1[b], "resnetfc_forward: NULL parameter (block %d)", b);
        {   // h = h + lin_z.b(z)
            GemmNT g;
            g.name = "gemm_generic_linz";
            set_segments(g, cfg, Z, tile_mask);
            g.W = net->w_z[b]; g.ldw = SCENERF_D_LATENT;
            g.M = M; g.N = H; g.bias = net->b_z[b];
            g.res = cur; g.ldres = H; g.res_f32 = 1;
            g.out = nxt; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
            float* t = cur; cur = nxt; nxt = t;
        }
        {   // net = fc_0(relu(h))
            GemmNT g;
            g.name = "gemm_generic_fc0";
            g.A1 = cur; g.lda1 = H; g.K1 = H; g.relu1 = 1;
            g.W = net->w_fc0[b]; g.ldw = H;
            g.M = M; g.N = H; g.bias = net->b_fc0[b];
            g.out = n_buf; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
        {   // h = h + fc_1(relu(net))
            GemmNT g;
            g.name = "gemm_generic_fc1";
            g.A1 = n_buf; g.lda1 = H; g.K1 = H; g.relu1 = 1;
            g.W = net->w_fc1[b]; g.ldw = H;
            g.M = M; g.N = H; g.bias = net->b_fc1[b];
            g.res = cur; g.ldres = H; g.res_f32 = 1;
            g.out = nxt; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
            float* t = cur; cur = nxt; nxt = t;
        }
    }
    GemmNT g;   // out = lin_out(relu(h)), the output columns zero-padded to a multiple of 8 by the caller
    g.name = "gemm_generic_lin_out";
    g.A1 = cur; g.lda1 = H; g.K1 = H; g.relu1 = 1;
    g.W = net->w_out; g.ldw = H;
    g.M = M; g.N = net->d_out_pad; g.bias = net->b_out;
    g.out = logits; g.ldout = net->d_out_pad; g.out_f32 = 1;
    return launch_gemm_nt(0, g, s);
}

// ---- the same net, trainable (round 6): the forward that saves what the backward reads, and the backward as one fp32-MFMA GEMM per
// gradient (scenerf_hip.h).  resnetfc.py:133-164 and its autograd.
static int resnetfc_check(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const char* who) {
    SRF_CHECK(cfg && net, "%s: NULL argument", who);
    SRF_CHECK(net->n_blocks >= 1 && net->n_blocks <= SCENERF_RESNETFC_MAX_BLOCKS, "%s: n_blocks=%d (1..%d)", who, net->n_blocks,
              SCENERF_RESNETFC_MAX_BLOCKS);
    SRF_CHECK(net->d_hidden >= 16 && net->d_hidden % 16 == 0 && net->d_out_pad >= 8 && net->d_out_pad % 8 == 0,
              "%s: d_hidden=%d must be a multiple of 16, d_out_pad=%d of 8", who, net->d_hidden, net->d_out_pad);
    SRF_CHECK(net->w_in && net->b_in && net->w_out && net->b_out, "%s: NULL parameter", who);
    for (int b = 0; b < net->n_blocks; ++b)
        SRF_CHECK(net->w_z[b] && net->b_z[b] && net->w_fc0[b] && net->b_fc0[b] && net->w_fc1[b] && net->b_fc1[b], "%s: NULL parameter (block %d)", who, b);
    return 0;
}

int scenerf_hip_resnetfc_forward_train(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const float* xenc, const float* Z,
                                       const uint8_t* tile_mask, int M, const scenerf_resnetfc_acts* acts, float* h_scratch,
                                       float* logits, scenerf_stream_t stream) {
    if (int e = resnetfc_check(cfg, net, "resnetfc_forward_train")) return e;
    SRF_CHECK(xenc && Z && tile_mask && acts && h_scratch && logits && M > 0 && acts->h_fin, "resnetfc_forward_train: NULL argument");
    hipStream_t s = as_stream(stream);
    const int H = net->d_hidden, nb = net->n_blocks;
    // h = lin_in(x) lands where the first block's residual is read from: the scratch for one block, h_fin's storage otherwise is free
    // until the last block writes it -- the running h alternates between the scratch and h_fin so that the LAST block's output is h_fin
    float* cur = (nb % 2) ? h_scratch : acts->h_fin;
    {
        GemmNT g;
        g.name = "gemm_generic_lin_in";
        g.A1 = xenc; g.lda1 = SCENERF_D_XENC; g.K1 = SCENERF_D_XENC;
        g.W = net->w_in; g.ldw = SCENERF_D_XENC;
        g.M = M; g.N = H; g.bias = net->b_in;
        g.out = cur; g.ldout = H; g.out_f32 = 1;
        if (int e = launch_gemm_nt(0, g, s)) return e;
    }
    for (int b = 0; b < nb; ++b) {
        SRF_CHECK(acts->hz[b] && acts->n[b], "resnetfc_forward_train: NULL activation buffer (block %d)", b);
        {   // hz[b] = h + lin_z.b(z)
            GemmNT g;
            g.name = "gemm_generic_linz";
            set_segments(g, cfg, Z, tile_mask);
            g.W = net->w_z[b]; g.ldw = SCENERF_D_LATENT;
            g.M = M; g.N = H; g.bias = net->b_z[b];
            g.res = cur; g.ldres = H; g.res_f32 = 1;
            g.out = acts->hz[b]; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
        {   // n[b] = fc_0(relu(hz[b]))
            GemmNT g;
            g.name = "gemm_generic_fc0";
            g.A1 = acts->hz[b]; g.lda1 = H; g.K1 = H; g.relu1 = 1;
            g.W = net->w_fc0[b]; g.ldw = H;
            g.M = M; g.N = H; g.bias = net->b_fc0[b];
            g.out = acts->n[b]; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
        {   // h = hz[b] + fc_1(relu(n[b]))
            float* nxt = cur == h_scratch ? acts->h_fin : h_scratch;
            GemmNT g;
            g.name = "gemm_generic_fc1";
            g.A1 = acts->n[b]; g.lda1 = H; g.K1 = H; g.relu1 = 1;
            g.W = net->w_fc1[b]; g.ldw = H;
            g.M = M; g.N = H; g.bias = net->b_fc1[b];
            g.res = acts->hz[b]; g.ldres = H; g.res_f32 = 1;
            g.out = nxt; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
            cur = nxt;
        }
    }
    SRF_CHECK(cur == acts->h_fin, "resnetfc_forward_train: internal buffer parity");
    GemmNT g;
    g.name = "gemm_generic_lin_out";
    g.A1 = cur; g.lda1 = H; g.K1 = H; g.relu1 = 1;
    g.W = net->w_out; g.ldw = H;
    g.M = M; g.N = net->d_out_pad; g.bias = net->b_out;
    g.out = logits; g.ldout = net->d_out_pad; g.out_f32 = 1;
    return launch_gemm_nt(0, g, s);
}

int scenerf_hip_resnetfc_backward(const scenerf_cfg* cfg, const scenerf_resnetfc* net, const scenerf_resnetfc_t* nt,
                                  const scenerf_resnetfc_grads* gr, const float* xenc, const float* Z, const uint8_t* tile_mask,
                                  const int32_t* tap_texel, const float* tap_weight, int M, const scenerf_resnetfc_acts* acts,
                                  const float* dlog16, float* dhz, float* dh, float* dn, float* const gmaps_hwc[SCENERF_N_SCALES],
                                  scenerf_stream_t stream) {
    if (int e = resnetfc_check(cfg, net, "resnetfc_backward")) return e;
    SRF_CHECK(nt && gr && xenc && Z && tile_mask && acts && dlog16 && dhz && dh && dn && M > 0 && acts->h_fin && nt->w_out_t,
              "resnetfc_backward: NULL argument");
    SRF_CHECK(!gmaps_hwc || (tap_texel && tap_weight), "resnetfc_backward: taps missing");
    SRF_CHECK(gr->w_in && gr->b_in && gr->w_z && gr->w_out && gr->b_out, "resnetfc_backward: NULL gradient buffer");
    hipStream_t s = as_stream(stream);
    const int H = net->d_hidden, nb = net->n_blocks, LD = nb * H;
    {   // lin_out: dW_out += dlog^T relu(h_fin), db_out += colsum(dlog)
        GemmTN t;
        t.name = "gemm_generic_wgrad_out";
        t.D = dlog16; t.ldd = 16; t.A = acts->h_fin; t.lda = H; t.relu_a = 1;
        t.M = M; t.N = 16; t.K = H; t.out = gr->w_out; t.ldo = H; t.colsum = gr->b_out;
        t.allow_tr = 0;
        if (int e = launch_gemm_tn(0, t, s)) return e;
    }
    {   // dh = (dlog W_out) * [h_fin > 0]
        GemmNT g;
        g.name = "gemm_generic_dgrad_out";
        g.A1 = dlog16; g.lda1 = 16; g.K1 = 16;
        g.W = nt->w_out_t; g.ldw = 16;
        g.M = M; g.N = H;
        g.maskp = acts->h_fin; g.ldmask = H;
        g.out = dh; g.ldout = H; g.out_f32 = 1;
        if (int e = launch_gemm_nt(0, g, s)) return e;
    }
    const float* dcur = dh;     // gradient w.r.t. the current block's output
    int ldcur = H;
    for (int b = nb - 1; b >= 0; --b) {
        SRF_CHECK(acts->hz[b] && acts->n[b] && nt->w_fc0_t[b] && nt->w_fc1_t[b] && gr->w_fc0[b] && gr->b_fc0[b] && gr->w_fc1[b] && gr->b_fc1[b],
                  "resnetfc_backward: NULL buffer (block %d)", b);
        {   // dW1 += dh^T relu(n[b]), db1 += colsum(dh)
            GemmTN t;
            t.name = "gemm_generic_wgrad_fc1";
            t.D = dcur; t.ldd = ldcur; t.A = acts->n[b]; t.lda = H; t.relu_a = 1;
            t.M = M; t.N = H; t.K = H; t.out = gr->w_fc1[b]; t.ldo = H; t.colsum = gr->b_fc1[b];
            t.allow_tr = 0;
            if (int e = launch_gemm_tn(0, t, s)) return e;
        }
        {   // dn = (dh W1) * [n[b] > 0]
            GemmNT g;
            g.name = "gemm_generic_dgrad_fc1";
            g.A1 = dcur; g.lda1 = ldcur; g.K1 = H;
            g.W = nt->w_fc1_t[b]; g.ldw = H;
            g.M = M; g.N = H;
            g.maskp = acts->n[b]; g.ldmask = H;
            g.out = dn; g.ldout = H; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
        {   // dW0 += dn^T relu(hz[b]), db0 += colsum(dn)
            GemmTN t;
            t.name = "gemm_generic_wgrad_fc0";
            t.D = dn; t.ldd = H; t.A = acts->hz[b]; t.lda = H; t.relu_a = 1;
            t.M = M; t.N = H; t.K = H; t.out = gr->w_fc0[b]; t.ldo = H; t.colsum = gr->b_fc0[b];
            t.allow_tr = 0;
            if (int e = launch_gemm_tn(0, t, s)) return e;
        }
        {   // dhz[b] = dh + (dn W0) * [hz[b] > 0]   (column block b of dhz)
            GemmNT g;
            g.name = "gemm_generic_dgrad_fc0";
            g.A1 = dn; g.lda1 = H; g.K1 = H;
            g.W = nt->w_fc0_t[b]; g.ldw = H;
            g.M = M; g.N = H;
            g.maskp = acts->hz[b]; g.ldmask = H;
            g.res2 = dcur; g.ldres2 = ldcur;
            g.out = dhz + (size_t)b * H; g.ldout = LD; g.out_f32 = 1;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
        dcur = dhz + (size_t)b * H;
        ldcur = LD;
    }
    // lin_z: dWz[b] += dhz[b]^T z, per pyramid level (row tiles without that level skipped: they multiply exact zeros)
    int off = 0;
    for (int sc = 0; sc < SCENERF_N_SCALES; ++sc) {
        const int c = cfg->map_C[sc];
        if (c > 0) {
            GemmTN t;
            t.name = "gemm_generic_wgrad_linz";
            t.D = dhz; t.ldd = LD;
            t.A = Z + off; t.lda = SCENERF_D_LATENT;
            t.M = M; t.N = LD; t.K = c;
            t.tile_mask = tile_mask; t.skip_bit = sc;
            t.out = gr->w_z + off; t.ldo = SCENERF_D_LATENT;
            t.allow_tr = 0;
            if (int e = launch_gemm_tn(0, t, s)) return e;
        }
        off += c;
    }
    {   // lin_in: dW_in += dhz[0]^T x, db_in += colsum(dhz[0])
        GemmTN t;
        t.name = "gemm_generic_wgrad_in";
        t.D = dhz; t.ldd = LD; t.A = xenc; t.lda = SCENERF_D_XENC;
        t.M = M; t.N = H; t.K = SCENERF_D_XENC; t.out = gr->w_in; t.ldo = SCENERF_D_XENC; t.colsum = gr->b_in;
        t.allow_tr = 0;
        if (int e = launch_gemm_tn(0, t, s)) return e;
    }
    if (gmaps_hwc) {   // dz = dhz Wz scattered through the forward's taps: one launch for all levels (the GEMM family's scatter epilogue)
        GemmNT g;
        g.name = "gemm_generic_dfeat_scatter";
        g.A1 = dhz; g.lda1 = LD; g.K1 = LD;
        g.ldw = LD;
        g.M = M;
        g.tile_mask = tile_mask;
        g.tap_texel = tap_texel; g.tap_weight = tap_weight;
        bool any = false;
        int t = 0;
        for (int sc = 0; sc < 5; ++sc) {
            SRF_CHECK(nt->w_z_t[sc] || !gmaps_hwc[sc], "resnetfc_backward: w_z_t[%d] missing", sc);
            g.ms_t0[sc] = t;
            g.ms_C[sc] = cfg->map_C[sc];
            g.ms_W[sc] = nt->w_z_t[sc];
            g.ms_gmap[sc] = gmaps_hwc[sc];
            if (cfg->map_chw[sc] == 1) { g.ms_st[sc] = 1; g.ms_sc[sc] = (long)cfg->map_H[sc] * cfg->map_W[sc]; }
            any = any || gmaps_hwc[sc];
            t += cdiv(cfg->map_C[sc], 128);
        }
        if (any) {
            g.ms_t0[5] = t;
            g.ms_n = 5;
            g.N = SCENERF_D_LATENT;
            g.scatter_scale = 0;
            if (int e = launch_gemm_nt(0, g, s)) return e;
        }
    }
    return 0;
}

int scenerf_hip_mlp_backward(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const scenerf_mlp_grads* g_,
                             const void* Z, const float* xenc, const uint8_t* tile_mask, const int32_t* tap_texel,
                             const float* tap_weight, int M, const scenerf_mlp_acts* a, const float* d_logits, void* dH,
                             void* dN, float* const gmaps_hwc[SCENERF_N_SCALES], scenerf_stream_t stream) {
    SRF_CHECK(cfg && w && g_ && Z && tile_mask && a && d_logits && dH && dN && M > 0 && (xenc || cfg->precision == 1), "mlp_backward: NULL argument");
    SRF_CHECK(!gmaps_hwc || (tap_texel && tap_weight), "mlp_backward: taps missing");
    // the dN scratch ([3][M][512] act) doubles as the lin_out partial-sum buffer before the block loop starts
    {
        const size_t blocks = (size_t)(cdiv(M, 128) > 2048 ? 2048 : cdiv(M, 128));
        SRF_CHECK((size_t)3 * M * SCENERF_D_HIDDEN * (cfg->precision ? 2 : 4) >= blocks * (4 * SCENERF_D_HIDDEN + 8) * 4,
                  "mlp_backward: dN scratch too small for the lin_out partial sums");
    }
    const int prec = cfg->precision;
    const bool head = w->d_out == 2;
    const size_t es = prec ? 2 : 4;
    hipStream_t s = as_stream(stream);
    const int LDH = 4 * SCENERF_D_HIDDEN;  // dH row: [dH0 | dH1 | dH2 | dH3]
    int kSegOff[5];
    for (int i = 0, off = 0; i < 5; ++i) { kSegOff[i] = off; off += cfg->map_C[i]; }
    auto dHcol = [&](int b) { return (void*)((char*)dH + (size_t)b * SCENERF_D_HIDDEN * es); };

    // Round 1 (per-layer GEMMs, 6.9 ms/step): neutral.  Round 3 (fused chain, 2.85 ms/step): the batched weight gradients on the side
    // stream beside the feature-map gradients on the caller's: -35 us per step in three same-process A/Bs (tools/ab_step.py
    // renderer.MAIN_WGRAD_OVERLAP) -- the renderer sets the flag for the radiance MLP.  (Also tried there: the coarser levels'
    // feature-gradient launch on a third stream beside the finest level's: +200 us, not kept.)
    const bool overlap = (cfg->flags & SCENERF_FLAG_WGRAD_OVERLAP) != 0;
    SideCtx* sc_ = overlap ? side_ctx(s) : nullptr;
    hipStream_t s2 = sc_ ? sc_->side : s;   // weight-gradient stream (== s when overlap is off)
    auto fork = [&]() -> int { return sc_ ? order_after(sc_, s, s2) : 0; };
    auto dNb = [&](int b) { return (void*)((char*)dN + (size_t)b * M * SCENERF_D_HIDDEN * es); };  // dN: [3][M][512]

    // bf16, enough rows: the whole dgrad chain (6 GEMMs) runs as ONE kernel (fused.hip / wide.hip); the weight-gradient GEMMs below then
    // only consume dH / dN
    const bool fused_chain = prec && w->w_stream && a->sign_bits && srf_use_fused(cfg, M) && !(cfg->flags & SCENERF_FLAG_NO_FUSED_BWD);
    const bool wide_chain = fused_chain && (cfg->flags & SCENERF_FLAG_WIDE_BWD) && (cdiv(M, 128) >= SRF_WIDE_MIN_BLOCKS || (cfg->flags & SCENERF_FLAG_WIDE_ANY_M));
    // the 128-row chain makes lin_out's input gradient dH3 = (d_logits W_out) * [H3 > 0] in its own prologue, from d_logits and H3's
    // sign bits (r03: as linout_bwd's output it was a 314 MB round trip, 97 us on the critical path of a KITTI step); lin_out's
    // weight / bias gradients are then all linout_bwd is asked for: 157 MB of H3 to read instead of a 314 MB round trip
    const bool dh3_in_chain = wide_chain && !(cfg->flags & SCENERF_FLAG_WIDE_BWD_STAGED);
    const int allow_tr = (cfg->flags & SCENERF_FLAG_NO_WGRAD_TR) ? 0 : 1;
    // Two-call form (round 6: SCENERF_FLAG_BWD_CHAIN_ONLY / _GRADS_ONLY, fused chain only): the caller may put a stream dependency between
    // the dgrad chain and the weight / feature-map gradients -- the renderer orders the radiance MLP's gradient phase behind the gaussian
    // head's whole backward, whose small kernels run beside the chain on another stream: when the chain got faster than they are (r06),
    // the batched weight-gradient launch (one workgroup per CU, 128 KiB of LDS each) met the head's last kernels AND the feature-gradient
    // launch on the CUs it needs, its last workgroups were placed only as those drained, and the launch took 1,036 us instead of 675.
    // Without a fused chain the layers' dgrad and weight-gradient GEMMs interleave: the CHAIN_ONLY call then does everything and the
    // GRADS_ONLY call nothing.
    const bool chain_only = (cfg->flags & SCENERF_FLAG_BWD_CHAIN_ONLY) != 0 && fused_chain;
    const bool grads_only = (cfg->flags & SCENERF_FLAG_BWD_GRADS_ONLY) != 0;
    SRF_CHECK(!((cfg->flags & SCENERF_FLAG_BWD_CHAIN_ONLY) && grads_only), "mlp_backward: CHAIN_ONLY and GRADS_ONLY are two calls, not one");
    if (grads_only && !fused_chain) return 0;
    if (grads_only) {
        if (int e = fork()) return e;
    }
    // lin_out backward -> dH3, dw_out, db_out
    if (grads_only) {
    } else if (dh3_in_chain) {
        // (measured, r04_b: the same reduction queued beside or behind the chain, on a side stream of equal or of lowest priority, with a
        // scratch buffer of its own -- nothing can share a CU with a block of the chain (all of its LDS and registers), so its workgroups
        // displace chain blocks by what they save in front of it: chain 537 -> 577 / 611 us, step time unchanged.  In stream order.)
        if (int e = launch_linout_bwd<bf16_t>(w->d_out, a->H[3], w->w_out, d_logits, M, nullptr, 0, g_->w_out, g_->b_out, (float*)dN, s)) return e;
    } else if (prec) {
        if (int e = launch_linout_bwd<bf16_t>(w->d_out, a->H[3], w->w_out, d_logits, M, dHcol(3), LDH, g_->w_out, g_->b_out, (float*)dN, s)) return e;
    } else {
        if (int e = launch_linout_bwd<float>(w->d_out, a->H[3], w->w_out, d_logits, M, dHcol(3), LDH, g_->w_out, g_->b_out, (float*)dN, s)) return e;
    }
    if (!dh3_in_chain && !grads_only) {
        if (int e = fork()) return e;
    }
    // (r04: the feature-map gradients launched in FRONT of the side stream's weight gradients -- so that, as the first-captured successor
    // of the chain kernel, they keep its queue in a replayed graph -- cost 36 us: the batched weight-gradient launch is one round of 240
    // long-lived workgroups, and behind dfeat's 1,200 it gets its CUs late and staggered: 742 against 593 us.  Weight gradients first.)
    if (fused_chain && !grads_only) {
        if (int e = wide_chain ? launch_mlp_bwd_wide(cfg, w, M, a, dH, dN, dh3_in_chain ? d_logits : nullptr, s) : launch_mlp_bwd_fused(cfg, w, M, a, dH, dN, s)) return e;
        if (chain_only) return 0;
        if (int e = fork()) return e;
    }
    GemmTN wg[8];   // the six fc weight gradients (+ lin_z, below): batched into one launch when the chain kernel already produced every dH / dN
    int nwg = 0;
    for (int b = 2; b >= 0; --b) {
        {   // [side] dW1_b += dH_{b+1}^T relu(N_b)
            GemmTN t;
            t.name = head ? "gemm_wgrad_fc1/g" : "gemm_wgrad_fc1";
            t.D = dHcol(b + 1); t.ldd = LDH; t.A = a->Nn[b]; t.lda = SCENERF_D_HIDDEN; t.relu_a = 1;
            t.M = M; t.N = SCENERF_D_HIDDEN; t.K = SCENERF_D_HIDDEN; t.out = g_->w_fc1[b]; t.ldo = SCENERF_D_HIDDEN;
            t.colsum = g_->b_fc1[b];  // fc_1.b.bias gradient = column sums of dH_{b+1}
            t.allow_tr = allow_tr;
            if (fused_chain && wgrad_tr_applicable(t, W_BATCH_MIN_ROWS)) { t.name = head ? "gemm_wgrad_fc/g" : "gemm_wgrad_fc"; wg[nwg++] = t; }
            else if (int e = launch_gemm_tn(prec, t, s2)) return e;
        }
        if (!fused_chain) {   // dN_b = (dH_{b+1} @ W1_b) * [N_b > 0]
            GemmNT g;
            g.name = head ? "gemm_dgrad_fc1/g" : "gemm_dgrad_fc1";
            g.A1 = dHcol(b + 1); g.lda1 = LDH; g.K1 = SCENERF_D_HIDDEN;
            g.W = w->w_fc1_t[b]; g.ldw = SCENERF_D_HIDDEN;
            g.M = M; g.N = SCENERF_D_HIDDEN;
            g.maskp = a->Nn[b]; g.ldmask = SCENERF_D_HIDDEN;
            g.out = dNb(b); g.ldout = SCENERF_D_HIDDEN;
            if (int e = launch_gemm_nt(prec, g, s)) return e;
        }
        if (!fused_chain) {
            if (int e = fork()) return e;
        }
        {   // [side] dW0_b += dN_b^T relu(H_b);  db0_b = column sums of dN_b
            GemmTN t;
            t.name = head ? "gemm_wgrad_fc0/g" : "gemm_wgrad_fc0";
            t.D = dNb(b); t.ldd = SCENERF_D_HIDDEN; t.A = a->H[b]; t.lda = SCENERF_D_HIDDEN; t.relu_a = 1;
            t.M = M; t.N = SCENERF_D_HIDDEN; t.K = SCENERF_D_HIDDEN; t.out = g_->w_fc0[b]; t.ldo = SCENERF_D_HIDDEN;
            t.colsum = g_->b_fc0[b];
            t.allow_tr = allow_tr;
            if (fused_chain && wgrad_tr_applicable(t, W_BATCH_MIN_ROWS)) { t.name = head ? "gemm_wgrad_fc/g" : "gemm_wgrad_fc"; wg[nwg++] = t; }
            else if (int e = launch_gemm_tn(prec, t, s2)) return e;
        }
        if (!fused_chain) {   // dH_b = dH_{b+1} + (dN_b @ W0_b) * [H_b > 0]
            GemmNT g;
            g.name = head ? "gemm_dgrad_fc0/g" : "gemm_dgrad_fc0";
            g.A1 = dNb(b); g.lda1 = SCENERF_D_HIDDEN; g.K1 = SCENERF_D_HIDDEN;
            g.W = w->w_fc0_t[b]; g.ldw = SCENERF_D_HIDDEN;
            g.M = M; g.N = SCENERF_D_HIDDEN;
            g.maskp = a->H[b]; g.ldmask = SCENERF_D_HIDDEN;
            g.res2 = dHcol(b + 1); g.ldres2 = LDH;
            g.out = dHcol(b); g.ldout = LDH;
            if (int e = launch_gemm_nt(prec, g, s)) return e;
        }
        if (!fused_chain) {
            if (int e = fork()) return e;
        }
    }
    // lin_z: dWz[:, slice_s] += dH[:, 0:1536]^T Z[:, slice_s].  In the batched launch the first 256 columns of Z -- the two finest
    // scales, which (nearly) every row block touches, and the head of the third -- ride along as a seventh problem without row
    // skipping (the gather writes exact zeros there for the row tiles that miss a scale: SCENERF_Z_DENSE_COLS); the per-scale
    // launches below then start at that column
    int linz_done = 0;
    if (nwg && SCENERF_Z_DENSE_COLS % 256 == 0) {
        GemmTN t;
        t.name = head ? "gemm_wgrad_fc/g" : "gemm_wgrad_fc";
        t.D = dH; t.ldd = LDH;
        t.A = Z; t.lda = SCENERF_D_LATENT;
        t.M = M; t.N = 3 * SCENERF_D_HIDDEN; t.K = SCENERF_Z_DENSE_COLS;
        t.out = g_->w_z; t.ldo = SCENERF_D_LATENT;
        t.allow_tr = allow_tr;
        if (wgrad_tr_applicable(t, W_BATCH_MIN_ROWS)) { wg[nwg++] = t; linz_done = SCENERF_Z_DENSE_COLS; }
    }
    // lin_in: dWin += dH0^T x as an eighth problem of the same launch (bf16): its K = 48 is padded to one 256-column tile -- the A tile
    // reads on into the following rows of the split encoding [M][144] (finite garbage, one row of slack behind the buffer), the
    // output tile's columns 48..255 land in the scratch part of w_in's rows (SCENERF_WIN_LD).  40 GFLOP of padding on the matrix
    // cores against a separate skinny GEMM that ran at 2 TB/s (76-82 us, r02/r03 profiles)
    int linin_done = 0;
    if (nwg && nwg < 8 && prec) {
        GemmTN t;
        t.name = head ? "gemm_wgrad_fc/g" : "gemm_wgrad_fc";
        t.D = dH; t.ldd = LDH;
        t.A = a->h0pre; t.lda = 3 * SCENERF_D_XENC;
        t.M = M; t.N = SCENERF_D_HIDDEN; t.K = SCENERF_WIN_LD; t.out = g_->w_in; t.ldo = SCENERF_WIN_LD;
        t.colsum = g_->b_in;
        t.allow_tr = allow_tr;
        if (wgrad_tr_applicable(t, W_BATCH_MIN_ROWS)) { wg[nwg++] = t; linin_done = 1; }
    }
    if (nwg) {
        if (int e = launch_wgrad_tr_batch(wg, nwg, s2)) return e;
    }
    // [side] the remaining (scale, column) ranges, row-tiles without scale s skipped
    for (int sc = 0; sc < 5; ++sc) {
        const int c0 = kSegOff[sc] > linz_done ? kSegOff[sc] : linz_done, c1 = kSegOff[sc] + cfg->map_C[sc];
        if (c1 <= c0) continue;
        GemmTN t;
        t.name = head ? "gemm_wgrad_linz/g" : "gemm_wgrad_linz";
        t.D = dH; t.ldd = LDH;
        t.A = (const char*)Z + (size_t)c0 * es; t.lda = SCENERF_D_LATENT;
        t.M = M; t.N = 3 * SCENERF_D_HIDDEN; t.K = c1 - c0;
        t.tile_mask = tile_mask; t.skip_bit = sc;
        t.out = g_->w_z + c0; t.ldo = SCENERF_D_LATENT;
        t.allow_tr = allow_tr;
        if (int e = launch_gemm_tn(prec, t, s2)) return e;
    }
    // [side] dWin += dH0^T xenc   (bf16 mode: the hi part of the split encoding kept in h0pre is bf16(xenc))
    if (!linin_done) {
        GemmTN t;
        t.name = head ? "gemm_wgrad_lin_in/g" : "gemm_wgrad_lin_in";
        t.D = dH; t.ldd = LDH;
        if (prec) {
            t.A = a->h0pre; t.lda = 3 * SCENERF_D_XENC;
        } else {
            t.A = xenc; t.lda = SCENERF_D_XENC;
        }
        t.M = M; t.N = SCENERF_D_HIDDEN; t.K = SCENERF_D_XENC; t.out = g_->w_in; t.ldo = SCENERF_WIN_LD;
        t.colsum = g_->b_in;  // lin_in.bias gradient = column sums of dH_0
        t.allow_tr = allow_tr;
        if (int e = launch_gemm_tn(prec, t, s2)) return e;
    }
    // lin_z.b.bias is added at the same place as lin_in.bias (b=0) / fc_1.(b-1).bias: same column sums of dH_b
    // (one 6-block kernel: three asynchronous device-to-device copies were three ~10 us blit launches per pass)
    copy3_kernel<<<3 * SCENERF_D_HIDDEN / 256 + (g_->w_in_dense ? cdiv(SCENERF_D_HIDDEN * 42, 256) : 0), 256, 0, s2>>>(
        g_->b_z, g_->b_in, g_->b_fc1[0], g_->b_fc1[1], g_->w_in_dense, g_->w_in);
    SRF_LAUNCH_CHECK("copy3_kernel");
    if (gmaps_hwc) {
        // the batched weight-gradient launch on s2 is ONE round of workgroups with 128 KiB of LDS each: it must find the CUs empty.  The
        // feature-gradient workgroups (two per CU, 1,200+ of them) launched at the same moment on this stream take the CUs first or
        // share them out, a weight-gradient workgroup then only fits when BOTH of a CU's feature-gradient workgroups have left -- and the
        // dispatcher refills the holes with more of those: the launch took 1,036-1,048 us instead of 675 (profiles/r06_h_*, r06_i_*: r05 got
        // away with a 6-us lead that the faster chain kernel of r06 no longer left).  A one-wave kernel that sleeps a few microseconds in
        // front of the feature gradients gives the other launch its head start, replayed graph or not.
        if (sc_ && nwg && srf_dfeat_delay_us() > 0 && !head) {
            srf_delay_kernel<<<1, 64, 0, s>>>(srf_dfeat_delay_us());
            SRF_LAUNCH_CHECK("srf_delay_kernel");
        }
        if (int e = feature_grads(cfg, w, tile_mask, tap_texel, tap_weight, M, dH, gmaps_hwc, s)) return e;
    }
    if (sc_) {  // join: the caller's stream continues only after the weight-gradient stream has drained
        if (int e = order_after(sc_, s2, s)) return e;
    }
    return 0;
}

int scenerf_hip_test_gemm_nt(int precision, const void* A, const void* W, const float* bias, int M, int N, int K, int relu_a,
                             int tile, float* C, scenerf_stream_t stream) {
    GemmNT g;
    g.name = "test_gemm_nt";
    g.force_tile = tile;
    g.A1 = A; g.lda1 = K; g.K1 = K; g.relu1 = relu_a;
    g.W = W; g.ldw = K; g.M = M; g.N = N; g.bias = bias;
    g.out = C; g.ldout = N; g.out_f32 = 1;
    return launch_gemm_nt(precision, g, as_stream(stream));
}

int scenerf_hip_test_chunk_table(const scenerf_cfg* cfg, int kind, int32_t* out, int cap) {
    if (!(cfg && out && (kind == 0 || kind == 2))) { srf_set_error("test_chunk_table: bad arguments (kind 0 = fused.hip, 2 = wide.hip)"); return -1; }
    std::vector<int> tab;
    if (int e = kind == 0 ? fused_table_build(cfg, tab) : wide_table_build(cfg, tab)) return -e;
    if ((int)tab.size() > cap) { srf_set_error("test_chunk_table: output buffer too small (%d ints needed)", (int)tab.size()); return -2; }
    for (size_t i = 0; i < tab.size(); ++i) out[i] = tab[i];
    return (int)tab.size();
}

int scenerf_hip_test_gemm_tn(int precision, const void* D, const void* A, int M, int N, int K, int relu_a, float* C,
                             float* colsum, scenerf_stream_t stream) {
    GemmTN t;
    t.name = "test_gemm_tn";
    t.colsum = colsum;
    t.D = D; t.ldd = N; t.A = A; t.lda = K; t.relu_a = relu_a;
    t.M = M; t.N = N; t.K = K; t.out = C; t.ldo = K;
    return launch_gemm_tn(precision, t, as_stream(stream));
}

}  // extern "C"

// ================================================================================================ operand packing
// nn.Linear parameters (fp32, reference layout) -> MFMA operand layout of scenerf_mlp_weights, in TWO launches
// (previously ~60 small torch kernels per MLP per step: concatenations, casts, transposes).
struct PackJob {
    const float* src;
    void* dst;
    int rows, cols;        // destination region
    int src_ld, dst_ld;
    int transpose;         // dst[r][c] = src[c][r]
    int mode;              // 0: cast, 1: bf16 hi part, 2: bf16 lo part (x - hi)
    int valid_cols;        // non-transposed: columns >= valid_cols are zero (padding)
    int dst_f32;           // destination element type: 1 = fp32, 0 = act type of the launch
    int tile0;             // index of this job's first 64x64 tile
};
#define PACK_MAX_JOBS 40
struct PackTable {
    PackJob job[PACK_MAX_JOBS];
    int njobs;
    // the four hidden-layer bias vectors ride in the same launch as two more blocks behind the tiles (a launch of their own was the third
    // dependent launch in front of a training step's first GEMM): b_h[0] = lin_z.0.bias (+ lin_in.bias when lin_in is fused),
    // b_h[1] = fc_1.0.bias + lin_z.1.bias, b_h[2] = fc_1.1.bias + lin_z.2.bias, b_h[3] = fc_1.2.bias
    int bias_block0;              // first of the two bias blocks, or -1
    const float *a0, *c0, *a1, *z1, *a2, *z2, *a3;
    float *o0, *o1, *o2, *o3;
};

template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(PackTable tab) {
    __shared__ float tile[64][65];
    if (tab.bias_block0 >= 0 && (int)blockIdx.x >= tab.bias_block0) {
        const int i = ((int)blockIdx.x - tab.bias_block0) * 256 + threadIdx.x;
        if (i < SCENERF_D_HIDDEN) {
            tab.o0[i] = tab.a0[i] + (tab.c0 ? tab.c0[i] : 0.f);
            tab.o1[i] = tab.a1[i] + tab.z1[i];
            tab.o2[i] = tab.a2[i] + tab.z2[i];
            tab.o3[i] = tab.a3[i];
        }
        return;
    }
    int j = 0;
    while (j + 1 < tab.njobs && (int)blockIdx.x >= tab.job[j + 1].tile0) ++j;
    const PackJob J = tab.job[j];
    const int t = blockIdx.x - J.tile0;
    const int tiles_c = (J.cols + 63) / 64;
    const int r0 = (t / tiles_c) * 64, c0 = (t % tiles_c) * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    if (J.transpose) {
        // read src rows c0.. (coalesced along src columns = dst rows), write dst rows r0.. (coalesced along dst columns)
        for (int k = ty; k < 64; k += 4) {
            const int sc = r0 + tx, sr = c0 + k;   // src[sr][sc] -> dst[sc][sr]
            tile[k][tx] = (sr < J.cols && sc < J.rows) ? J.src[(size_t)sr * J.src_ld + sc] : 0.f;
        }
        __syncthreads();
    }
    for (int k = ty; k < 64; k += 4) {
        const int r = r0 + k, c = c0 + tx;
        if (r >= J.rows || c >= J.cols) continue;
        float v;
        if (J.transpose) v = tile[tx][k];
        else v = (c < J.valid_cols) ? J.src[(size_t)r * J.src_ld + c] : 0.f;
        if (J.mode) {
            const float hi = bf16_to_f32(f32_to_bf16(v));
            v = (J.mode == 1) ? hi : v - hi;
        }
        if (J.dst_f32) ((float*)J.dst)[(size_t)r * J.dst_ld + c] = v;
        else ActIO<T>::st(J.dst, (size_t)r * J.dst_ld + c, v);
    }
}

// w_stream (scenerf_hip.h): the seven forward operands re-tiled into 16 KiB blocks [512 rows][32 B] per 16 columns of K, the
// two 16-byte halves of row r swapped when (r >> 3) & 1 -- the LDS image fused.hip's fragment reads expect, so a streaming
// piece (1 KiB per wave) is contiguous in memory.  One thread per 16-byte half row.
#define STREAM_OPS 13
struct StreamSrc {
    const bf16_t* W[STREAM_OPS];
    int ld[STREAM_OPS];
    int block0[STREAM_OPS + 1];   // first block of each operand; the last entry = total
    int first, last;              // this launch writes blocks [first, last)
};
__global__ __launch_bounds__(256) void pack_stream_kernel(StreamSrc t, uint4* __restrict__ dst) {
    const int idx = (blockIdx.x + 4 * t.first) * 256 + threadIdx.x;
    const int blk = idx >> 10;
    if (blk >= t.last) return;
    int l = 0;
    while (blk >= t.block0[l + 1]) ++l;
    const int r = (idx & 1023) >> 1, ps = idx & 1;
    const int k = (blk - t.block0[l]) * 16 + 8 * (ps ^ ((r >> 3) & 1));
    dst[idx] = *(const uint4*)(t.W[l] + (size_t)r * t.ld[l] + k);
}

extern "C" int scenerf_hip_mlp_pack(const scenerf_cfg* cfg, const scenerf_mlp_params* P, const scenerf_mlp_weights* W,
                                    scenerf_stream_t stream) {
    SRF_CHECK(cfg && P && W, "mlp_pack: NULL argument");
    SRF_CHECK(P->d_out == 4 || P->d_out == 2, "mlp_pack: d_out must be 4 or 2");
    const int prec = cfg->precision;
    hipStream_t s = as_stream(stream);
    const int H = SCENERF_D_HIDDEN, L = SCENERF_D_LATENT, X = SCENERF_D_XENC;
    const size_t es = prec ? 2 : 4;
    PackTable tab;
    tab.njobs = 0;
    int tiles = 0;
    // Two-call form (bf16 with the streaming layout; SCENERF_FLAG_PACK_FORWARD, then SCENERF_FLAG_PACK_REST): the first call packs what a
    // FORWARD pass reads (the untransposed operands, their streaming blocks, the biases), the second the rest (transposed operands
    // and their streaming blocks, the per-level W_z^T, the gradient sink's zeroes) -- the gaussian head's forward is the first MFMA
    // kernel of a training step and waited ~30 us for a pack of which it needs a third (r04 trace)
    const bool split_ok = prec && W->w_stream;
    const bool only_fwd = split_ok && (cfg->flags & SCENERF_FLAG_PACK_FORWARD), only_rest = split_ok && (cfg->flags & SCENERF_FLAG_PACK_REST);
    SRF_CHECK(!(only_fwd && only_rest), "mlp_pack: SCENERF_FLAG_PACK_FORWARD and _REST are two calls");
    bool fwd_job = true;   // class of the jobs being added
    auto add = [&](const float* src, void* dst, int rows, int cols, int src_ld, int dst_ld, int transpose, int mode, int valid_cols,
                   int dst_f32) {
        if ((only_fwd && !fwd_job) || (only_rest && fwd_job)) return;
        if (tab.njobs >= PACK_MAX_JOBS) { ++tab.njobs; return; }
        PackJob& J = tab.job[tab.njobs++];
        J.src = src; J.dst = dst; J.rows = rows; J.cols = cols; J.src_ld = src_ld; J.dst_ld = dst_ld;
        J.transpose = transpose; J.mode = mode; J.valid_cols = valid_cols; J.dst_f32 = dst_f32; J.tile0 = tiles;
        tiles += cdiv(rows, 64) * cdiv(cols, 64);
    };
    auto at = [&](const void* base, size_t elem_off) { return (void*)((char*)base + elem_off * es); };
    for (int i = 0; i < 3; ++i) SRF_CHECK(P->fc0_w[i] && P->fc1_w[i] && P->linz_w[i] && P->fc1_b[i] && P->linz_b[i], "mlp_pack: NULL parameter");
    SRF_CHECK(P->lin_in_w && P->lin_in_b && W->w_in && W->w_h[0] && W->b_h[0], "mlp_pack: NULL parameter");
    // lin_in: fp32 copy zero-padded 42 -> 48 (used by the fp32 path and by nothing else in bf16 mode)
    add(P->lin_in_w, (void*)W->w_in, H, X, 42, X, 0, 0, 42, 1);
    // first hidden GEMM
    if (prec) {
        const int ld0 = 3 * X + L;
        add(P->lin_in_w, at(W->w_h[0], 0), H, X, 42, ld0, 0, 1, 42, 0);        // w_hi  (against x_hi)
        add(P->lin_in_w, at(W->w_h[0], X), H, X, 42, ld0, 0, 1, 42, 0);        // w_hi  (against x_lo)
        add(P->lin_in_w, at(W->w_h[0], 2 * X), H, X, 42, ld0, 0, 2, 42, 0);    // w_lo  (against x_hi)
        add(P->linz_w[0], at(W->w_h[0], 3 * X), H, L, L, ld0, 0, 0, L, 0);
    } else {
        add(P->linz_w[0], (void*)W->w_h[0], H, L, L, L, 0, 0, L, 0);
    }
    for (int b = 0; b < 3; ++b) {
        const int ld = b < 2 ? H + L : H;
        add(P->fc1_w[b], (void*)W->w_h[b + 1], H, H, H, ld, 0, 0, H, 0);
        if (b < 2) add(P->linz_w[b + 1], at(W->w_h[b + 1], H), H, L, L, ld, 0, 0, L, 0);
        add(P->fc0_w[b], (void*)W->w_fc0[b], H, H, H, H, 0, 0, H, 0);
        fwd_job = false;
        add(P->fc0_w[b], (void*)W->w_fc0_t[b], H, H, H, H, 1, 0, H, 0);
        add(P->fc1_w[b], (void*)W->w_fc1_t[b], H, H, H, H, 1, 0, H, 0);
        fwd_job = true;
    }
    fwd_job = false;
    int off = 0;
    for (int sc = 0; sc < 5; ++sc) {   // w_z_t[s][c][b*512 + n] = lin_z.b.weight[n][off_s + c]
        for (int b = 0; b < 3; ++b)
            add(P->linz_w[b] + off, at(W->w_z_t[sc], (size_t)b * H), cfg->map_C[sc], H, L, 3 * H, 1, 0, H, 0);
        off += cfg->map_C[sc];
    }
    if (W->clear && W->clear_floats > 0) {   // zero jobs: no source (valid_cols = 0 -> every element is padding)
        const int64_t cols = 4096, full = W->clear_floats / cols, rem = W->clear_floats - full * cols;
        SRF_CHECK(full < (1 << 30), "mlp_pack: clear buffer too large");
        if (full > 0) add(nullptr, W->clear, (int)full, (int)cols, 0, (int)cols, 0, 0, 0, 1);
        if (rem > 0) add(nullptr, W->clear + full * cols, 1, (int)rem, 0, (int)rem, 0, 0, 0, 1);
    }
    SRF_CHECK(tab.njobs <= PACK_MAX_JOBS, "mlp_pack: job table overflow");
    tab.bias_block0 = only_rest ? -1 : tiles;
    tab.a0 = P->linz_b[0]; tab.c0 = prec ? P->lin_in_b : nullptr; tab.a1 = P->fc1_b[0]; tab.z1 = P->linz_b[1]; tab.a2 = P->fc1_b[1];
    tab.z2 = P->linz_b[2]; tab.a3 = P->fc1_b[2];
    tab.o0 = (float*)W->b_h[0]; tab.o1 = (float*)W->b_h[1]; tab.o2 = (float*)W->b_h[2]; tab.o3 = (float*)W->b_h[3];
    const int blocks = tiles + (only_rest ? 0 : 2);
    if (blocks > 0) {
        SrfLaunchScope ps(s, "mlp_pack", 0, 0);
        if (prec) pack_kernel<bf16_t><<<blocks, 256, 0, s>>>(tab);
        else pack_kernel<float><<<blocks, 256, 0, s>>>(tab);
        SRF_LAUNCH_CHECK("pack_kernel");
    }
    if (prec && W->w_stream) {
        StreamSrc t;
        const void* ops[STREAM_OPS] = {W->w_h[0], W->w_fc0[0], W->w_h[1], W->w_fc0[1], W->w_h[2], W->w_fc0[2], W->w_h[3],
                                       W->w_fc1_t[2], W->w_fc0_t[2], W->w_fc1_t[1], W->w_fc0_t[1], W->w_fc1_t[0], W->w_fc0_t[0]};
        const int lds_[STREAM_OPS] = {3 * X + L, H, H + L, H, H + L, H, H, H, H, H, H, H, H};
        int nb = 0;
        for (int i = 0; i < STREAM_OPS; ++i) {
            t.W[i] = (const bf16_t*)ops[i];
            t.ld[i] = lds_[i];
            t.block0[i] = nb;
            nb += lds_[i] / 16;
        }
        t.block0[STREAM_OPS] = nb;
        SRF_CHECK(nb == SCENERF_W_STREAM_BLOCKS, "mlp_pack: stream block count");
        t.first = only_rest ? t.block0[7] : 0;      // operands 0..6: the forward's, 7..12: the dgrad chain's
        t.last = only_fwd ? t.block0[7] : nb;
        SrfLaunchScope ps(s, "mlp_pack_stream", 0, (double)(t.last - t.first) * 32768);
        pack_stream_kernel<<<(t.last - t.first) * 4, 256, 0, s>>>(t, (uint4*)W->w_stream);
        SRF_LAUNCH_CHECK("pack_stream_kernel");
    }
    return 0;
}
