// Per-ray / per-sample kernels of the SceneRF hot path for gfx950 (wave64):
//   ray setup, point encoding (projection + spherical index + positional encoding), HWC feature gather,
//   gaussian sampling + in-LDS bitonic sort, wave-per-ray alpha compositing (fwd/bwd), RaySOM-KL (fwd),
//   sampler/KL backward, and the CHW<->HWC feature-map layout changes.
// Compiled with -ffp-contract=off: every fused multiply-add below is an explicit fmaf, so the rounding
// sequence is the one documented next to each formula (it follows the reference's eager torch ops).
#include "common.h"
#include "sphere_exact.h"

#define PI_F 3.14159265358979323846f
#define HALF_PI_F 1.57079632679489661923f

// k-ordered fma chain == what a BLAS sgemm micro-kernel does for a length-3/4 dot product (sphere_exact.h: checked bit for bit against
// torch's `K @ p` on the CPU by tests/test_sphere_exact.py)
#define dot3 srf_dot3
#define dot4 srf_dot4

// ------------------------------------------------------------------------------------------------ ray setup
// one thread per (ray, j): j < U writes dist_u; j == 0 also writes unit_dir / viewdir.
// rng (device uint64 {seed, calls, calls'} or NULL): with noise_u == NULL the uniform noise of utils.py:84 is made here (common.h: Philox
// on (element, call)).  The call counter is passed on in ping-pong fashion so that no launch reads a word another thread of the same
// launch writes: this kernel reads rng[1] and leaves rng[1] + 1 in rng[2]; gaussian_sample_sort_kernel reads rng[2] and copies it to
// rng[1] -- a replayed hipGraph therefore draws fresh noise on every replay.
__global__ void ray_setup_kernel(const float* __restrict__ pixels, const float* __restrict__ iK,
                                 const float* __restrict__ T, const float* __restrict__ lin_u,
                                 const float* __restrict__ noise_u, unsigned long long* __restrict__ rng, int R, int U, float step,
                                 float* __restrict__ unit_dir, float* __restrict__ viewdir, float* __restrict__ dist_u) {
    int W = U > 0 ? U : 1;
    int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long call = rng ? rng[1] : 0ull;
    if (rng && gid == 0) rng[2] = call + 1ull;
    if (gid >= R * W) return;
    int r = gid / W, j = gid - r * W;
    if (j < U) {
        float nu;
        if (noise_u) nu = noise_u[(size_t)r * U + j];
        else {
            uint32_t o[4];
            srf_philox((uint32_t)gid, (uint32_t)call, (uint32_t)(call >> 32) ^ 0x0u, rng[0], o);
            nu = srf_u01(o[0]);
        }
        dist_u[(size_t)r * U + j] = lin_u[j] + nu * step;  // utils.py:84-85
    }
    if (j == 0) {
        // utils.py:177-182 (unit dirs), :170 (un-normalised view direction): sphere_exact.h, torch-CPU's operation sequence
        srf_ray_dir(iK, T, pixels[2 * r], pixels[2 * r + 1], unit_dir + 3 * r, viewdir + 3 * r);
    }
}

// ------------------------------------------------------------------------------------------------ encode
struct SphereConsts {
    float v_min, v_fov, h_min, h_fov;
    int W, H;
};

// x3 (bf16 mode, may be NULL): the split-bf16 encoding [M][144] = [hi(48) | lo(48) | hi(48)] the first hidden GEMM consumes (mlp.hip:
// split_xenc_kernel -- same values, bit for bit), written from here so that the fp32 encoding never makes the round trip through HBM.
// FOUR threads per row (round 4): the 36 precise sines of a row's positional encoding are ~3,600 instructions -- with one thread per
// row the kernel was one long serial chain per thread (18-28 us for the gaussian head's 4,800 rows on 19 CUs, 22 us for a training
// chunk's 153,600: latency, not throughput) and 34 KB of straight-line code; thread (row, part) now evaluates the (frequency, phase)
// pairs 3 part .. 3 part + 2, i.e. nine sines.  Every thread of a row recomputes the row's point (30 instructions); part 0 writes the
// per-row outputs.  A block's 64 rows are staged in LDS (the fp32 encoding, then its split form) and leave as fully coalesced 16-byte
// stores (a row is 192 / 288 bytes: the block's rows are one contiguous range).
#define ENC_X3_ROW (3 * SCENERF_D_XENC * 2)          // 288 bytes
#define ENC_X3_LD (ENC_X3_ROW + 16)                   // LDS row stride (16-byte aligned, off the 32-bank period)
#define ENC_ROWS 64                                   // rows per 256-thread block
#define ENC_F32_LD (SCENERF_D_XENC + 4)               // fp32 staging row stride in floats (16-byte aligned rows)
__global__ __launch_bounds__(256) void encode_points_kernel(const float* __restrict__ dist, int dist_ray_stride, int ppr,
                                     const float* __restrict__ unit_dir, const float* __restrict__ viewdir,
                                     const float* __restrict__ K, const float* __restrict__ iK,
                                     const float* __restrict__ T, SphereConsts sc, int M,
                                     float* __restrict__ pts_out, int32_t* __restrict__ sphere_idx,
                                     float* __restrict__ xenc, bf16_t* __restrict__ x3) {
    __shared__ __attribute__((aligned(16))) float s_f32[ENC_ROWS * ENC_F32_LD];
    __shared__ __attribute__((aligned(16))) char s_x3[ENC_ROWS * ENC_X3_LD];
    const int row = threadIdx.x >> 2, part = threadIdx.x & 3;
    const int m0 = blockIdx.x * ENC_ROWS;
    const int m_raw = m0 + row;
    const bool live = m_raw < M;
    const int m = live ? m_raw : M - 1;   // (rows past M: computed on the last row, never stored)
    int r = m / ppr, j = m - r * ppr;
    float d = dist[(size_t)r * dist_ray_stride + j];
    // source-frame point = dist * unit_dir (utils.py:87 / 217), then T @ [p,1] (utils.py:161-166)
    float qv[3];
    srf_sample_point(T, unit_dir + 3 * r, d, qv);
    const float qx = qv[0], qy = qv[1], qz = qv[2];
    if (part == 0) {
        if (pts_out && live) {
            pts_out[3 * (size_t)m] = qx;
            pts_out[3 * (size_t)m + 1] = qy;
            pts_out[3 * (size_t)m + 2] = qz;
        }
        // cam_pts_2_pix (utils.py:298-315), then SphericalMapping.from_pixels at depth 1 (spherical_mapping.py:80-115): the operation
        // sequence of torch's CPU kernels, acos / atan2 included (sphere_exact.h) -- the rounded index is the reference's, bit for bit
        float u, v, ox, oy;
        srf_cam_pt_to_pix(K, qx, qy, qz, &u, &v);
        srf_sphere_consts sxc;
        sxc.v_min = sc.v_min; sxc.v_fov = sc.v_fov; sxc.h_min = sc.h_min; sxc.h_fov = sc.h_fov; sxc.W = sc.W; sxc.H = sc.H;
        srf_pix_to_sphere_f(iK, sxc, u, v, &ox, &oy);
        if (live) {
            sphere_idx[2 * (size_t)m] = srf_round_index(ox);
            sphere_idx[2 * (size_t)m + 1] = srf_round_index(oy);
        }
    }
    // PositionalEncoding pe.py:32-43: [x, sin(f0 x), sin(f0 x + pi/2), ...] then viewdir, zero pad to 48.  Columns of this thread:
    // 3 + 3 i + c for the (frequency, phase) pairs i = 3 part .. 3 part + 2 (pair i = frequency i / 2, phase i & 1); part 0 adds columns
    // 0..2 (the point), part 3 columns 39..47 (viewdir, padding)
    const float q[3] = {qx, qy, qz};
    float e[9];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int i = 3 * part + t;
        float f = PI_F;
        for (int k = 0; k < (i >> 1); ++k) f *= 2.f;      // (pi 2^k: the reference's freq_factor * 2 ** k, exact doublings)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float a = q[c] * f;  // addcmul: phase + x*f, product rounded first
            e[3 * t + c] = sinf((i & 1) ? HALF_PI_F + a : a);
        }
    }
    // the row's fp32 encoding into LDS, column by column ...
    float* const rowf = s_f32 + row * ENC_F32_LD;
#pragma unroll
    for (int t = 0; t < 9; ++t) rowf[3 + 9 * part + t] = e[t];
    if (part == 0) { rowf[0] = qx; rowf[1] = qy; rowf[2] = qz; }
    if (part == 3) {
        rowf[39] = viewdir[3 * r]; rowf[40] = viewdir[3 * r + 1]; rowf[41] = viewdir[3 * r + 2];
#pragma unroll
        for (int c = 42; c < SCENERF_D_XENC; ++c) rowf[c] = 0.f;
    }
    __syncthreads();
    const int nrows = min(M - m0, ENC_ROWS);
    if (xenc) {   // ... out as the block's contiguous [nrows][48] fp32 range, 16 bytes per thread and pass
        float* const g = xenc + (size_t)m0 * SCENERF_D_XENC;
        for (int v = threadIdx.x; v < nrows * (SCENERF_D_XENC / 4); v += 256) {
            const int rr = v / (SCENERF_D_XENC / 4), cc = v - rr * (SCENERF_D_XENC / 4);
            *(float4*)(g + (size_t)rr * SCENERF_D_XENC + 4 * cc) = *(const float4*)(s_f32 + rr * ENC_F32_LD + 4 * cc);
        }
    }
    if (x3) {
        // one row of slack behind [M][144]: lin_in's weight gradient reads the rows in 256-column tiles (scenerf_hip.h: h0pre) -- zeroed here
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x < ENC_X3_ROW / 4) ((uint32_t*)((char*)x3 + (size_t)M * ENC_X3_ROW))[threadIdx.x] = 0u;
        // ... and split: thread (row, part) takes columns 12 part .. 12 part + 11.  x = hi + lo, hi = bf16(x), lo = bf16(x - hi): mlp.hip
        // split_xenc_kernel
        char* const rowp = s_x3 + row * ENC_X3_LD;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const int c = 12 * part + 2 * t;
            const float v0 = rowf[c], v1 = rowf[c + 1];
            const uint32_t hi = pack_bf16x2(v0, v1);
            const uint32_t lo = pack_bf16x2(v0 - bf16lo(hi), v1 - bf16hi(hi));
            *(uint32_t*)(rowp + c * 2) = hi;
            *(uint32_t*)(rowp + SCENERF_D_XENC * 2 + c * 2) = lo;
            *(uint32_t*)(rowp + SCENERF_D_XENC * 4 + c * 2) = hi;
        }
        __syncthreads();
        char* const g = (char*)x3 + (size_t)m0 * ENC_X3_ROW;
        for (int v = threadIdx.x; v < nrows * (ENC_X3_ROW / 16); v += 256) {
            const int rr = v / (ENC_X3_ROW / 16), cc = v - rr * (ENC_X3_ROW / 16);
            *(uint4*)(g + (size_t)rr * ENC_X3_ROW + 16 * cc) = *(const uint4*)(s_x3 + rr * ENC_X3_LD + 16 * cc);
        }
    }
}

// ------------------------------------------------------------------------------------------------ gather
struct GatherConsts {
    int C[5], Hm[5], Wm[5], Hd[5], Wd[5], off[5];
    int chw[5];   // scenerf_cfg.map_chw: 1 = read from the caller's fp32 (C,H,W) map, 2 = from the caller's fp32 (H,W,C) map, 0 = (H,W,C) act copy
};

template <typename T> struct Vec16;  // 16-byte vector of T
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static inline void load(const float* p, float* v) {
        float4 t = *(const float4*)p;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static inline void store(float* p, const float* v) { *(float4*)p = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ static inline void load(const bf16_t* p, float* v) {
        uint4 t = *(const uint4*)p;
        v[0] = bf16lo(t.x); v[1] = bf16hi(t.x); v[2] = bf16lo(t.y); v[3] = bf16hi(t.y);
        v[4] = bf16lo(t.z); v[5] = bf16hi(t.z); v[6] = bf16lo(t.w); v[7] = bf16hi(t.w);
    }
    __device__ static inline void store(bf16_t* p, const float* v) {
        *(uint4*)p = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                pack_bf16x2(v[6], v[7]));
    }
};

struct MapPtrs {
    const void* p[5];
};

// one 256-thread block per 128-row tile (gridDim.y == 1), or per (tile, pyramid level) when gridDim.y == 5: small launches -- the
// gaussian head's 38 tiles on 256 CUs took 42 us, each block walking all the levels its tile touches -- spread their levels over
// five times as many blocks; every block still derives the tile's full level mask (arithmetic only), block y == 0 publishes it.
// Phase 1: taps + validity per (row, scale); phase 2: blend.
// GU = items (row, 16-byte channel chunk) whose taps a thread requests together in the blend: 3 for the latency-bound launches (a
// training chunk: 1,200 tiles), 1 for the throughput-bound ones (an inference chunk's 8,192 tiles: the 96 extra registers of GU = 3 cost
// more occupancy than the loads in flight buy: 286 -> 364 us per launch, r03)
template <typename T, int GU>
__global__ __launch_bounds__(256) void gather_kernel(MapPtrs maps, GatherConsts gc, const int32_t* __restrict__ sphere_idx,
                                                     int M, T* __restrict__ Z, uint8_t* __restrict__ tile_mask,
                                                     int32_t* __restrict__ tap_texel, float* __restrict__ tap_weight) {
    __shared__ int s_tex[SCENERF_TILE_ROWS][5][4];
    __shared__ float s_w[SCENERF_TILE_ROWS][5][4];
    __shared__ unsigned s_mask;
    __shared__ uint8_t s_rowbits[SCENERF_TILE_ROWS];   // per row: scales with at least one in-range tap
    __shared__ uint8_t s_list[5][SCENERF_TILE_ROWS];   // per scale: the rows with taps, compacted (direct (C,H,W) scales)
    __shared__ int s_cnt[5];
    const int tile = blockIdx.x;
    const int tid = threadIdx.x;
    const int s_lo = gridDim.y == 1 ? 0 : (int)blockIdx.y, s_hi = gridDim.y == 1 ? 5 : (int)blockIdx.y + 1;   // levels of this block
    if (tid == 0) s_mask = 0u;
    if (tid < 5) s_cnt[tid] = 0;
    __syncthreads();
    if (tid < SCENERF_TILE_ROWS) {
        int m = tile * SCENERF_TILE_ROWS + tid;
        unsigned bits = 0u;
        int ix = 0, iy = 0;
        if (m < M) {
            ix = sphere_idx[2 * (size_t)m];
            iy = sphere_idx[2 * (size_t)m + 1];
        }
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            int tex[4] = {-1, -1, -1, -1};
            float w[4] = {0.f, 0.f, 0.f, 0.f};
            if (m < M) {
                // utils.py:237: idx / div * 2 - 1 ; ATen grid_sampler (align_corners=False): (g+1)*size/2 - 0.5
                float gx = (float)ix / (float)gc.Wd[s] * 2.f - 1.f;
                float gy = (float)iy / (float)gc.Hd[s] * 2.f - 1.f;
                float fx = (gx + 1.f) * ((float)gc.Wm[s] * 0.5f) - 0.5f;
                float fy = (gy + 1.f) * ((float)gc.Hm[s] * 0.5f) - 0.5f;
                if (fx > -2.f && fx < (float)gc.Wm[s] + 1.f && fy > -2.f && fy < (float)gc.Hm[s] + 1.f) {
                    float x0f = floorf(fx), y0f = floorf(fy);
                    float wx = fx - x0f, ex = 1.f - wx, wy = fy - y0f, sy = 1.f - wy;
                    int x0 = (int)x0f, y0 = (int)y0f;
                    float ww[4] = {sy * ex, sy * wx, wy * ex, wy * wx};  // nw, ne, sw, se
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        int xx = x0 + (t & 1), yy = y0 + (t >> 1);
                        if (xx >= 0 && xx < gc.Wm[s] && yy >= 0 && yy < gc.Hm[s]) {
                            tex[t] = yy * gc.Wm[s] + xx;
                            w[t] = ww[t];
                            bits |= 1u << s;
                        }
                    }
                }
                if (s >= s_lo && s < s_hi) {   // (one 16-byte store each: the four taps of a (row, level) are consecutive)
                    size_t o = ((size_t)m * 5 + s) * 4;
                    *(int4*)(tap_texel + o) = make_int4(tex[0], tex[1], tex[2], tex[3]);
                    *(float4*)(tap_weight + o) = make_float4(w[0], w[1], w[2], w[3]);
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                s_tex[tid][s][t] = tex[t];
                s_w[tid][s][t] = w[t];
            }
        }
        s_rowbits[tid] = (uint8_t)bits;
        for (int sc = 0; sc < 5; ++sc)
            if (gc.chw[sc] == 1 && ((bits >> sc) & 1u)) s_list[sc][atomicAdd(&s_cnt[sc], 1)] = (uint8_t)tid;
        if (bits) atomicOr(&s_mask, bits);
    }
    __syncthreads();
    const unsigned mask = s_mask;
    if (tid == 0 && blockIdx.y == 0) tile_mask[tile] = (uint8_t)mask;
    constexpr int VN = Vec16<T>::N;
    for (int s = s_lo; s < s_hi; ++s) {
        if (!(mask & (1u << s))) {
            // a scale this tile does not touch: its columns are normally never read (every consumer skips them by tile_mask) and stay
            // unwritten -- except the first SCENERF_Z_DENSE_COLS columns, which the batched lin_z weight gradient reads for every row
            // (mlp.hip): those get exact zeros
            const int c1 = min(gc.off[s] + gc.C[s], SCENERF_Z_DENSE_COLS);
            const int nzc = (c1 - gc.off[s]) / VN;
            if (nzc > 0) {
                float z[VN];
#pragma unroll
                for (int e = 0; e < VN; ++e) z[e] = 0.f;
                for (int it = tid; it < SCENERF_TILE_ROWS * nzc; it += 256) {
                    const int row = it / nzc, ch = it - row * nzc;
                    Vec16<T>::store(Z + ((size_t)tile * SCENERF_TILE_ROWS + row) * SCENERF_D_LATENT + gc.off[s] + ch * VN, z);
                }
            }
            continue;
        }
        const int C = gc.C[s];
        const int chunks = C / VN;
        if (gc.chw[s] == 1) {
            // this scale was not converted: blend straight from the caller's fp32 (C,H,W) map.  Rows without a tap (nearly all of
            // them at the coarse scales) get their zeros from the vector loop; a row with taps spreads its C x 4 strided loads
            // over the whole workgroup -- they are independent cache lines, so parallelism is what hides them
            const float* chw = (const float*)maps.p[s];
            const size_t HW = (size_t)gc.Hm[s] * gc.Wm[s];
            {
                float z[VN];
#pragma unroll
                for (int e = 0; e < VN; ++e) z[e] = 0.f;
                for (int it = tid; it < SCENERF_TILE_ROWS * chunks; it += 256) {
                    const int row = it / chunks, ch = it - row * chunks;
                    if (!((s_rowbits[row] >> s) & 1u))
                        Vec16<T>::store(Z + ((size_t)tile * SCENERF_TILE_ROWS + row) * SCENERF_D_LATENT + gc.off[s] + ch * VN, z);
                }
            }
            // all (row with taps, channel) pairs of the tile in one flat loop: every load of the workgroup is independent
            const int nv = s_cnt[s];
            for (int idx = tid; idx < nv * C; idx += 256) {
                const int row = s_list[s][idx / C], c = idx % C;
                float acc = 0.f;
                bool first = true;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int tx = s_tex[row][s][t];
                    if (tx >= 0) {
                        const float v = chw[(size_t)c * HW + tx], w = s_w[row][s][t];
                        acc = first ? v * w : fmaf(v, w, acc);
                        first = false;
                    }
                }
                ActIO<T>::st(Z + ((size_t)tile * SCENERF_TILE_ROWS + row) * SCENERF_D_LATENT + gc.off[s], c, acc);
            }
            continue;
        }
        // (H,W,C) maps: the act copy, or the caller's fp32 map read in place (scenerf_cfg.map_chw == 2: the blend sees the unrounded
        // features, Z is rounded once; fp32 mode reads exactly that through the generic branch).  Item = (row, 16-byte channel chunk of
        // Z).  The taps of GU items are requested together, branch-free -- an out-of-range tap reads texel 0 and is replaced by zeros
        // after the load; its weight is 0 -- so a thread has 4 GU independent loads in flight instead of one dependent wait per tap
        // (r03: the guarded form ran the main gather at 1.1 TB/s of requests, one memory latency after the other).  Same arithmetic:
        // the first in-range tap used to be a plain product, fmaf(v, w, 0) rounds identically; a skipped tap is fmaf(0, 0, acc) = acc.
        const int items = SCENERF_TILE_ROWS * chunks;
        const bool inplace32 = gc.chw[s] == 2 && sizeof(T) != 4;
        for (int it0 = tid; it0 < items; it0 += 256 * GU) {
            float v[GU][4][VN], w[GU][4];
            int row[GU], ch[GU];
            bool ok[GU][4];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int it = min(it0 + 256 * u, items - 1);
                row[u] = it / chunks;
                ch[u] = it - row[u] * chunks;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int tx = s_tex[row[u]][s][t];
                    ok[u][t] = tx >= 0;
                    w[u][t] = s_w[row[u]][s][t];
                    const size_t o = (size_t)max(tx, 0) * C + ch[u] * VN;
                    if (inplace32) {
                        const float* mapf = (const float*)maps.p[s];
#pragma unroll
                        for (int q = 0; q < VN / 4; ++q) {
                            const float4 f = *(const float4*)(mapf + o + 4 * q);
                            v[u][t][4 * q] = f.x; v[u][t][4 * q + 1] = f.y; v[u][t][4 * q + 2] = f.z; v[u][t][4 * q + 3] = f.w;
                        }
                    } else {
                        Vec16<T>::load((const T*)maps.p[s] + o, v[u][t]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                float acc[VN];
#pragma unroll
                for (int e = 0; e < VN; ++e) acc[e] = 0.f;
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int e = 0; e < VN; ++e) acc[e] = fmaf(ok[u][t] ? v[u][t][e] : 0.f, w[u][t], acc[e]);
                if (it0 + 256 * u < items)
                    Vec16<T>::store(Z + ((size_t)tile * SCENERF_TILE_ROWS + row[u]) * SCENERF_D_LATENT + gc.off[s] + ch[u] * VN, acc);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ sampler + sort
// one wave per ray; N <= 512 keys sorted with a bitonic network in LDS (stable via (key, index) compare).
// rng != NULL: the normal noise of utils.py:208-211 is made here and WRITTEN to noise_g (the sampler's backward reads it back)
__global__ __launch_bounds__(64) void gaussian_sample_sort_kernel(
    const float* __restrict__ offsets, const float* __restrict__ anchors, const float* __restrict__ dist_u,
    float* __restrict__ noise_g, unsigned long long* __restrict__ rng, const float* __restrict__ unit_dir, int R, int U, int G, int P, int N, int NP,
    float base_std, float floor_, float* __restrict__ gmeans, float* __restrict__ gstds,
    float* __restrict__ dist_sorted, float* __restrict__ z_sorted, int32_t* __restrict__ perm) {
    __shared__ float s_key[SCENERF_MAX_SAMPLES];
    __shared__ int s_idx[SCENERF_MAX_SAMPLES];
    __shared__ float s_mean[SCENERF_MAX_GAUSSIANS], s_std[SCENERF_MAX_GAUSSIANS];
    const int r = blockIdx.x, lane = threadIdx.x;
    const unsigned long long call = rng ? rng[2] - 1ull : 0ull;      // (ray_setup_kernel left calls + 1 there)
    if (rng && r == 0 && lane == 0) rng[1] = call + 1ull;
    if (lane < G) {
        float o0 = offsets[((size_t)r * G + lane) * 2], o1 = offsets[((size_t)r * G + lane) * 2 + 1];
        float mean = fmaxf(anchors[lane] + o0, 0.f) + floor_;  // scenerf.py:588-592
        float sd = fmaxf(o1 + base_std, 0.f) + floor_;         // scenerf.py:593-594
        s_mean[lane] = mean;
        s_std[lane] = sd;
        gmeans[(size_t)r * G + lane] = mean;
        gstds[(size_t)r * G + lane] = sd;
    }
    __syncthreads();
    for (int j = lane; j < NP; j += 64) {
        float key = __builtin_inff();
        if (j < N) {
            if (j < U) {
                key = dist_u[(size_t)r * U + j];
            } else {
                int jj = j - U, g = jj / P;
                float nz;
                if (rng) {
                    uint32_t o[4];
                    srf_philox((uint32_t)(r * G * P + jj), (uint32_t)call, (uint32_t)(call >> 32) ^ 0x80000000u, rng[0], o);
                    nz = srf_normal(o[0], o[1]);
                    noise_g[(size_t)r * G * P + jj] = nz;
                } else {
                    nz = noise_g[(size_t)r * G * P + jj];
                }
                float d = s_mean[g] + nz * s_std[g];  // utils.py:213
                key = d < 0.1f ? 0.1f : d;                                           // utils.py:214
            }
        }
        s_key[j] = key;
        s_idx[j] = j;
    }
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < NP / 2; t += 64) {
                int i = ((t / j) * 2 * j) + (t % j);
                int l = i + j;
                bool up = (i & k) == 0;
                float ka = s_key[i], kb = s_key[l];
                int ia = s_idx[i], ib = s_idx[l];
                bool gt = (ka > kb) || (ka == kb && ia > ib);
                if (gt == up) {
                    s_key[i] = kb; s_key[l] = ka;
                    s_idx[i] = ib; s_idx[l] = ia;
                }
            }
            __syncthreads();
        }
    }
    const float uz = unit_dir[3 * r + 2];
    for (int j = lane; j < N; j += 64) {
        float d = s_key[j];
        dist_sorted[(size_t)r * N + j] = d;
        z_sorted[(size_t)r * N + j] = d * uz;  // cam_pts[:, :, 2], utils.py:159 / 219
        perm[(size_t)r * N + j] = s_idx[j];
    }
}

// ------------------------------------------------------------------------------------------------ compositing
__device__ static inline float softplus_m1(float x) {  // nn.Softplus(beta=1)(x - 1), threshold 20
    float y = x - 1.f;
    return y > 20.f ? y : log1pf(expf(y));
}
__device__ static inline float sigmoidf(float x) { return 1.f / (1.f + expf(-x)); }

// wave-wide exclusive product scan of one value per lane
__device__ static inline float wave_excl_prod(float p, int lane) {
    float incl = p;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(incl, o, WAVE);
        if (lane >= o) incl *= t;
    }
    float ex = __shfl_up(incl, 1, WAVE);
    return lane == 0 ? 1.f : ex;
}
// wave-wide exclusive suffix sum (sum over lanes > lane)
__device__ static inline float wave_excl_suffix_sum(float v, int lane) {
    float incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_down(incl, o, WAVE);
        if (lane + o < 64) incl += t;
    }
    float ex = __shfl_down(incl, 1, WAVE);
    return lane == 63 ? 0.f : ex;
}

// wave-wide inclusive->total helpers
__device__ static inline float wave_total_prod_from_excl(float excl, float own) { return __shfl(excl * own, 63, WAVE); }

// One wave per ray; the ray's N samples are processed in C = ceil(N/64) segments of 64, lane l owning sample 64 c + l of
// segment c, so every global access of a segment is one fully coalesced 256-byte (fp32) / 1-KiB (float4 logits) run.
// Per-ray state lives in VGPRs; the in-ray transmittance product is a 6-step __shfl_up scan per segment with the
// running product carried across segments (wave-uniform).  4 rays (waves) per 256-thread block.
template <int C>
__global__ __launch_bounds__(256) void composite_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ dist,
                                                            const float* __restrict__ zv, int R, int N,
                                                            float* __restrict__ densities, float* __restrict__ alphas,
                                                            float* __restrict__ weights, float* __restrict__ depth,
                                                            float* __restrict__ color, float* __restrict__ closest,
                                                            float* __restrict__ w_at, int32_t* __restrict__ closest_idx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t base = (size_t)r * N;
    float z[C], w[C];
    float4 lg[C];
    float d[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {   // issue all loads first
        const int i = c * 64 + lane;
        const bool ok = i < N;
        lg[c] = ok ? *(const float4*)(logits + (base + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dd = ok ? dist[base + i] : 0.f;
        d[c] = dd < 0.f ? 0.f : dd;  // scenerf.py:707
        z[c] = ok ? zv[base + i] : 0.f;
    }
    float carryT = 1.f, carry_d = 0.f;
    float sd = 0.f, sr = 0.f, sgc = 0.f, sb = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        const float sg = softplus_m1(lg[c].w);
        const float cr = sigmoidf(lg[c].x), cg = sigmoidf(lg[c].y), cb = sigmoidf(lg[c].z);
        float before = __shfl_up(d[c], 1, WAVE);
        if (lane == 0) before = carry_d;
        const float delta = (i == 0) ? d[c] : d[c] - before;             // scenerf.py:708-710
        const float a = ok ? 1.f - expf(-delta * sg) : 0.f;
        const float sfac = 1.f - a + 1e-10f;
        const float excl = wave_excl_prod(sfac, lane);                   // cumprod, scenerf.py:718-721
        const float Ti = carryT * excl;
        w[c] = a * Ti;                                                   // scenerf.py:723
        carryT *= wave_total_prod_from_excl(excl, sfac);
        carry_d = __shfl(d[c], 63, WAVE);
        sd += w[c] * z[c];
        sr += w[c] * cr;
        sgc += w[c] * cg;
        sb += w[c] * cb;
        if (ok) {
            densities[base + i] = sg;
            alphas[base + i] = a;
            weights[base + i] = w[c];
        }
    }
    sd = wave_sum(sd);
    sr = wave_sum(sr);
    sgc = wave_sum(sgc);
    sb = wave_sum(sb);
    // closest sample to the rendered depth (first minimum), scenerf.py:730-735
    float best = __builtin_inff();
    int bi = 0x7fffffff;
    float bw = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        if (i < N) {
            const float a = fabsf(sd - z[c]);
            if (a < best) { best = a; bi = i; bw = w[c]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(bi, o, WAVE);
        const float ow = __shfl_xor(bw, o, WAVE);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; bw = ow; }
    }
    if (lane == 0) {
        depth[r] = sd;
        color[3 * r] = sr;
        color[3 * r + 1] = sgc;
        color[3 * r + 2] = sb;
        closest[r] = best;
        w_at[r] = bw;
        closest_idx[r] = bi;
    }
}

// backward: recompute sigma/alpha/T/w in-wave (nothing but the inputs is re-read), then walk the segments in reverse
// with a carried suffix sum.  dL/dalpha_i = gw_i T_i - (sum_{j>i} gw_j w_j) / (1 - alpha_i + 1e-10).
template <int C>
__global__ __launch_bounds__(256) void composite_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ dist,
                                                            const float* __restrict__ zv, int R, int N,
                                                            const float* __restrict__ g_depth, const float* __restrict__ g_color,
                                                            const float* __restrict__ g_weights, const float* __restrict__ g_alphas,
                                                            const float* __restrict__ g_dens, const float* __restrict__ g_zvol,
                                                            float* __restrict__ d_logits, float* __restrict__ d_dist,
                                                            float* __restrict__ d_z) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t base = (size_t)r * N;
    float d[C], z[C], sg[C], al[C], w[C], cr[C], cg[C], cb[C], dl[C], Ti[C], o3[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        const float4 lg = ok ? *(const float4*)(logits + (base + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dd = ok ? dist[base + i] : 0.f;
        d[c] = dd < 0.f ? 0.f : dd;
        z[c] = ok ? zv[base + i] : 0.f;
        o3[c] = lg.w;
        sg[c] = softplus_m1(lg.w);
        cr[c] = sigmoidf(lg.x);
        cg[c] = sigmoidf(lg.y);
        cb[c] = sigmoidf(lg.z);
    }
    const float gd = g_depth[r];
    const float gcr = g_color[3 * r], gcg = g_color[3 * r + 1], gcb = g_color[3 * r + 2];
    float carryT = 1.f, carry_d = 0.f;
    float gw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        float before = __shfl_up(d[c], 1, WAVE);
        if (lane == 0) before = carry_d;
        dl[c] = (i == 0) ? d[c] : d[c] - before;
        al[c] = ok ? 1.f - expf(-dl[c] * sg[c]) : 0.f;
        const float sfac = 1.f - al[c] + 1e-10f;
        const float excl = wave_excl_prod(sfac, lane);
        Ti[c] = carryT * excl;
        w[c] = al[c] * Ti[c];
        carryT *= wave_total_prod_from_excl(excl, sfac);
        carry_d = __shfl(d[c], 63, WAVE);
        float g = gd * z[c] + gcr * cr[c] + gcg * cg[c] + gcb * cb[c];
        if (g_weights && ok) g += g_weights[base + i];
        gw[c] = ok ? g : 0.f;
    }
    float carryS = 0.f;        // sum of gw*w over all later segments
    float next_first = 0.f;    // gdelta of the first sample of the next segment (i + 1 for lane 63)
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        const float v = gw[c] * w[c];
        const float S = carryS + wave_excl_suffix_sum(v, lane);
        carryS += wave_sum(v);
        const float s_i = 1.f - al[c] + 1e-10f;
        float ga = gw[c] * Ti[c] - S / s_i;
        if (g_alphas && ok) ga += g_alphas[base + i];
        const float one_m = expf(-dl[c] * sg[c]);  // = 1 - alpha
        float gs = ga * dl[c] * one_m;
        if (g_dens && ok) gs += g_dens[base + i];
        const float gdelta = ok ? ga * sg[c] * one_m : 0.f;
        // delta_i = d_i - d_{i-1}  =>  dL/dd_i = gdelta_i - gdelta_{i+1}
        float after = __shfl_down(gdelta, 1, WAVE);
        if (lane == 63) after = next_first;
        next_first = __shfl(gdelta, 0, WAVE);
        if (ok) {
            const float y = o3[c] - 1.f;
            const float dsig = y > 20.f ? 1.f : sigmoidf(y);  // softplus'
            float4 o;
            o.x = gcr * w[c] * cr[c] * (1.f - cr[c]);
            o.y = gcg * w[c] * cg[c] * (1.f - cg[c]);
            o.z = gcb * w[c] * cb[c] * (1.f - cb[c]);
            o.w = gs * dsig;
            *(float4*)(d_logits + (base + i) * 4) = o;
            float gz = gd * w[c];
            if (g_zvol) gz += g_zvol[base + i];
            d_z[base + i] = gz;
            d_dist[base + i] = gdelta - ((i + 1 < N) ? after : 0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------ RaySOM
#define MAXG SCENERF_MAX_GAUSSIANS
// one wave per ray; lanes stride over the N samples.  ray_som_kl.py:10-87.  GM = compile-time bound of the gaussian count (4 for every
// configuration the reference ships, 8 = SCENERF_MAX_GAUSSIANS otherwise): with the bound at 8 a ray of 4 gaussians paid 8 expf and 8
// divisions per sample and pass -- the per-ray tail is latency (one wave walks a ray: 38 us at R = 1,200 whatever the chip could do)
template <int GM>
__global__ __launch_bounds__(256) void raysom_fwd_kernel(const float* __restrict__ gmeans, const float* __restrict__ gstds,
                                                         const float* __restrict__ dist, const float* __restrict__ alphas,
                                                         int R, int N, int G, float som_sigma, float kl_floor,
                                                         float* __restrict__ loss_kl, float* __restrict__ som_means,
                                                         float* __restrict__ som_vars, float* __restrict__ kl_saved,
                                                         uint8_t* __restrict__ bmu_out) {
    __shared__ float s_nb[4][GM][GM], s_p12[4][GM][GM];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wv;
    const bool active = r < R;
    float m[GM], s[GM], var[GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        bool ok = active && g < G;
        m[g] = ok ? gmeans[(size_t)r * G + g] : 0.f;
        s[g] = ok ? gstds[(size_t)r * G + g] : 1.f;
        var[g] = s[g] * s[g];
    }
    const float two_sig2 = (float)(2.0 * (double)som_sigma * (double)som_sigma);
    if (lane < GM * GM) {
        int c2 = lane / GM, c1 = lane % GM;
        float dm = m[0];  // select m[c2]-m[c1] without dynamic register indexing
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g == c2) a = m[g];
            if (g == c1) b = m[g];
        }
        dm = a - b;
        s_nb[wv][c2][c1] = (c2 < G && c1 < G) ? expf(-(dm * dm) / two_sig2) : 0.f;  // ray_som_kl.py:89-91
    }
    __syncthreads();
    if (lane < GM * GM) {
        int c2 = lane / GM, c1 = lane % GM;
        float sum = 0.f;
        for (int g = 0; g < G; ++g) sum += s_nb[wv][c2][g];
        s_p12[wv][c2][c1] = (c2 < G && c1 < G) ? s_nb[wv][c2][c1] / sum : 0.f;
    }
    __syncthreads();
    if (!active) return;
    const float sqrt2pi = 2.5066282746310002f;
    float sw[GM], swd[GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) { sw[g] = 0.f; swd[g] = 0.f; }
    // pass 1: weighted means
    for (int i = lane; i < N; i += 64) {
        float d = dist[(size_t)r * N + i];
        float dens = alphas[(size_t)r * N + i] + 1e-8f;
        float pz1[GM];
#pragma unroll
        for (int c = 0; c < GM; ++c) {
            float gap = fabsf(m[c] - d);
            float p = expf(-(gap * gap) / (2.f * var[c])) / (sqrt2pi * s[c]) + 1e-5f;
            pz1[c] = (c < G) ? p * dens + 1e-8f : 0.f;
        }
        float pbest = -1.f;
        int bmu = 0;
        for (int c2 = 0; c2 < G; ++c2) {
            float acc = 0.f;
#pragma unroll
            for (int c1 = 0; c1 < GM; ++c1)
                if (c1 < G) acc += pz1[c1] * s_p12[wv][c2][c1] + 1e-8f;
            if (acc > pbest) { pbest = acc; bmu = c2; }
        }
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g < G) {
                float wgt = s_nb[wv][g][bmu] * pz1[g] / pbest + 1e-5f;
                sw[g] += wgt;
                swd[g] += wgt * d;
            }
        }
    }
    float nm[GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        sw[g] = wave_sum(sw[g]);
        swd[g] = wave_sum(swd[g]);
        nm[g] = swd[g] / sw[g];
    }
    // pass 2: weighted variances around the new means (weights recomputed, not stored)
    float sv[GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) sv[g] = 0.f;
    for (int i = lane; i < N; i += 64) {
        float d = dist[(size_t)r * N + i];
        float dens = alphas[(size_t)r * N + i] + 1e-8f;
        float pz1[GM];
#pragma unroll
        for (int c = 0; c < GM; ++c) {
            float gap = fabsf(m[c] - d);
            float p = expf(-(gap * gap) / (2.f * var[c])) / (sqrt2pi * s[c]) + 1e-5f;
            pz1[c] = (c < G) ? p * dens + 1e-8f : 0.f;
        }
        float pbest = -1.f;
        int bmu = 0;
        for (int c2 = 0; c2 < G; ++c2) {
            float acc = 0.f;
#pragma unroll
            for (int c1 = 0; c1 < GM; ++c1)
                if (c1 < G) acc += pz1[c1] * s_p12[wv][c2][c1] + 1e-8f;
            if (acc > pbest) { pbest = acc; bmu = c2; }
        }
        if (bmu_out) bmu_out[(size_t)r * N + i] = (uint8_t)bmu;   // parity tests: the discrete choice of ray_som_kl.py:52
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g < G) {
                float wgt = s_nb[wv][g][bmu] * pz1[g] / pbest + 1e-5f;
                float e = d - nm[g];
                sv[g] += wgt * (e * e);
            }
        }
    }
    float klsum = 0.f;
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        sv[g] = wave_sum(sv[g]);
        if (g < G) {
            float nv = sv[g] / sw[g];
            float mean_diff = fabsf(m[g] - nm[g]);
            float std_diff = fabsf(sqrtf(var[g]) - sqrtf(nv));
            float mk = ((mean_diff > 0.1f) && (nv > 0.f) && (std_diff > 0.1f)) ? 1.f : 0.f;  // ray_som_kl.py:66-70
            float s2 = sqrtf(nv);
            if (s2 < kl_floor) s2 = kl_floor;  // ray_som_kl.py:83
            float dmm = m[g] - nm[g];
            float kl = logf(s2 / s[g] + 1e-8f) + (s[g] * s[g] + dmm * dmm) / (2.f * (s2 * s2)) - 0.5f;
            klsum += kl * mk;
            if (lane == 0) {
                som_means[(size_t)r * G + g] = nm[g];
                som_vars[(size_t)r * G + g] = nv;
                kl_saved[((size_t)r * G + g) * 3 + 0] = nm[g];
                kl_saved[((size_t)r * G + g) * 3 + 1] = s2;
                kl_saved[((size_t)r * G + g) * 3 + 2] = mk;
            }
        }
    }
    if (lane == 0) loss_kl[r] = klsum / (float)G;
}


// ------------------------------------------------------------------------------------------------ fused per-ray tail
// composite_fwd_kernel + raysom_fwd_kernel in ONE launch (scenerf.py:704-748 + ray_som_kl.py:10-87): the ray's sorted distances and
// the alphas the compositing just produced stay in the wave's registers for the SOM update -- one launch and one pass over
// (logits, dist, z) per ray instead of two launches and a re-read of dist / alphas.  Same arithmetic, statement for statement, as
// the two stage kernels (tests hold the outputs bit-identical to theirs).
template <int GM>
__device__ __forceinline__ void raysom_sample(const float d, const float dens, const float (&m)[GM], const float (&s)[GM],
                                              const float (&var)[GM], const int G, const float (*p12)[GM], float (&pz1)[GM],
                                              float& pbest, int& bmu) {
    const float sqrt2pi = 2.5066282746310002f;
#pragma unroll
    for (int c = 0; c < GM; ++c) {
        float gap = fabsf(m[c] - d);
        float p = expf(-(gap * gap) / (2.f * var[c])) / (sqrt2pi * s[c]) + 1e-5f;
        pz1[c] = (c < G) ? p * dens + 1e-8f : 0.f;
    }
    pbest = -1.f;
    bmu = 0;
    for (int c2 = 0; c2 < G; ++c2) {
        float acc = 0.f;
#pragma unroll
        for (int c1 = 0; c1 < GM; ++c1)
            if (c1 < G) acc += pz1[c1] * p12[c2][c1] + 1e-8f;
        if (acc > pbest) { pbest = acc; bmu = c2; }
    }
}

// SOM = false (round 5): the compositing alone -- no RaySOM tables, no barriers, no second and third pass over the registers -- for callers
// that do not ask for loss_kl / som_vars (full-frame inference, keys = depth / colour: the SOM update was computed and thrown away).
// densities / alphas / weights are written only where their pointer is non-NULL: a depth + colour render moves 20 N + 24 bytes per ray
// instead of 32 N + 24.
template <int C, int GM, bool SOM>
__global__ __launch_bounds__(256) void ray_tail_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ dist,
                                                           const float* __restrict__ zv, const float* __restrict__ gmeans,
                                                           const float* __restrict__ gstds, int R, int N, int G, float som_sigma,
                                                           float kl_floor, float* __restrict__ densities, float* __restrict__ alphas,
                                                           float* __restrict__ weights, float* __restrict__ depth,
                                                           float* __restrict__ color, float* __restrict__ closest,
                                                           float* __restrict__ w_at, int32_t* __restrict__ closest_idx,
                                                           float* __restrict__ loss_kl, float* __restrict__ som_means,
                                                           float* __restrict__ som_vars, float* __restrict__ kl_saved,
                                                           uint8_t* __restrict__ bmu_out) {
    __shared__ float s_nb[SOM ? 4 : 1][GM][GM], s_p12[SOM ? 4 : 1][GM][GM];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + wv;
    const bool active = r < R;
    // ---- RaySOM tables of this ray (ray_som_kl.py:30-38), before anything else: the barriers are block-wide
    float m[GM], s[GM], var[GM];
    if (SOM) {
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        bool ok = active && g < G;
        m[g] = ok ? gmeans[(size_t)r * G + g] : 0.f;
        s[g] = ok ? gstds[(size_t)r * G + g] : 1.f;
        var[g] = s[g] * s[g];
    }
    const float two_sig2 = (float)(2.0 * (double)som_sigma * (double)som_sigma);
    if (lane < GM * GM) {
        int c2 = lane / GM, c1 = lane % GM;
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int g = 0; g < GM; ++g) {
            if (g == c2) a = m[g];
            if (g == c1) b = m[g];
        }
        const float dm = a - b;
        s_nb[wv][c2][c1] = (c2 < G && c1 < G) ? expf(-(dm * dm) / two_sig2) : 0.f;
    }
    __syncthreads();
    if (lane < GM * GM) {
        int c2 = lane / GM, c1 = lane % GM;
        float sum = 0.f;
        for (int g = 0; g < G; ++g) sum += s_nb[wv][c2][g];
        s_p12[wv][c2][c1] = (c2 < G && c1 < G) ? s_nb[wv][c2][c1] / sum : 0.f;
    }
    __syncthreads();
    }
    if (!active) return;
    // ---- compositing (composite_fwd_kernel)
    const size_t base = (size_t)r * N;
    float z[C], w[C], d[C], al[C];
    float4 lg[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        lg[c] = ok ? *(const float4*)(logits + (base + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dd = ok ? dist[base + i] : 0.f;
        d[c] = dd < 0.f ? 0.f : dd;  // scenerf.py:707 (in place on the sorted tensor: RaySOM sees the clamped distances too)
        z[c] = ok ? zv[base + i] : 0.f;
    }
    float carryT = 1.f, carry_d = 0.f;
    float sd = 0.f, sr = 0.f, sgc = 0.f, sb = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        const float sg = softplus_m1(lg[c].w);
        const float cr = sigmoidf(lg[c].x), cg = sigmoidf(lg[c].y), cb = sigmoidf(lg[c].z);
        float before = __shfl_up(d[c], 1, WAVE);
        if (lane == 0) before = carry_d;
        const float delta = (i == 0) ? d[c] : d[c] - before;
        const float a = ok ? 1.f - expf(-delta * sg) : 0.f;
        al[c] = a;
        const float sfac = 1.f - a + 1e-10f;
        const float excl = wave_excl_prod(sfac, lane);
        const float Ti = carryT * excl;
        w[c] = a * Ti;
        carryT *= wave_total_prod_from_excl(excl, sfac);
        carry_d = __shfl(d[c], 63, WAVE);
        sd += w[c] * z[c];
        sr += w[c] * cr;
        sgc += w[c] * cg;
        sb += w[c] * cb;
        if (ok) {
            if (densities) densities[base + i] = sg;
            if (alphas) alphas[base + i] = a;
            if (weights) weights[base + i] = w[c];
        }
    }
    sd = wave_sum(sd);
    sr = wave_sum(sr);
    sgc = wave_sum(sgc);
    sb = wave_sum(sb);
    float best = __builtin_inff();
    int bi = 0x7fffffff;
    float bw = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        if (i < N) {
            const float a = fabsf(sd - z[c]);
            if (a < best) { best = a; bi = i; bw = w[c]; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, WAVE);
        const int oi = __shfl_xor(bi, o, WAVE);
        const float ow = __shfl_xor(bw, o, WAVE);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; bw = ow; }
    }
    if (lane == 0) {
        depth[r] = sd;
        color[3 * r] = sr;
        color[3 * r + 1] = sgc;
        color[3 * r + 2] = sb;
        closest[r] = best;
        w_at[r] = bw;
        closest_idx[r] = bi;
    }
    if (!SOM) return;
    // ---- RaySOM update + KL (raysom_fwd_kernel) on the registers' (distance, alpha)
    // (the per-sample update weights of pass 1 are kept for pass 2 -- C x G registers -- instead of being recomputed: the stage kernel
    // evaluates the same expressions twice, the values are identical)
    float sw[GM], swd[GM], wk[C][GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) { sw[g] = 0.f; swd[g] = 0.f; }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
#pragma unroll
        for (int g = 0; g < GM; ++g) wk[c][g] = 0.f;
        if (i < N) {
            float pz1[GM], pbest;
            int bmu;
            raysom_sample<GM>(d[c], al[c] + 1e-8f, m, s, var, G, s_p12[wv], pz1, pbest, bmu);
            if (bmu_out) bmu_out[base + i] = (uint8_t)bmu;
#pragma unroll
            for (int g = 0; g < GM; ++g) {
                if (g < G) {
                    float wgt = s_nb[wv][g][bmu] * pz1[g] / pbest + 1e-5f;
                    wk[c][g] = wgt;
                    sw[g] += wgt;
                    swd[g] += wgt * d[c];
                }
            }
        }
    }
    float nm[GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        sw[g] = wave_sum(sw[g]);
        swd[g] = wave_sum(swd[g]);
        nm[g] = swd[g] / sw[g];
    }
    float sv[GM];
#pragma unroll
    for (int g = 0; g < GM; ++g) sv[g] = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        if (i < N) {
#pragma unroll
            for (int g = 0; g < GM; ++g) {
                if (g < G) {
                    float e = d[c] - nm[g];
                    sv[g] += wk[c][g] * (e * e);
                }
            }
        }
    }
    float klsum = 0.f;
#pragma unroll
    for (int g = 0; g < GM; ++g) {
        sv[g] = wave_sum(sv[g]);
        if (g < G) {
            float nv = sv[g] / sw[g];
            float mean_diff = fabsf(m[g] - nm[g]);
            float std_diff = fabsf(sqrtf(var[g]) - sqrtf(nv));
            float mk = ((mean_diff > 0.1f) && (nv > 0.f) && (std_diff > 0.1f)) ? 1.f : 0.f;  // ray_som_kl.py:66-70
            float s2 = sqrtf(nv);
            if (s2 < kl_floor) s2 = kl_floor;  // ray_som_kl.py:83
            float dmm = m[g] - nm[g];
            float kl = logf(s2 / s[g] + 1e-8f) + (s[g] * s[g] + dmm * dmm) / (2.f * (s2 * s2)) - 0.5f;
            klsum += kl * mk;
            if (lane == 0) {
                som_means[(size_t)r * G + g] = nm[g];
                som_vars[(size_t)r * G + g] = nv;
                kl_saved[((size_t)r * G + g) * 3 + 0] = nm[g];
                kl_saved[((size_t)r * G + g) * 3 + 1] = s2;
                kl_saved[((size_t)r * G + g) * 3 + 2] = mk;
            }
        }
    }
    if (lane == 0) loss_kl[r] = klsum / (float)G;
}

// sampler + KL backward: one wave per ray, 4 rays per block.
__global__ __launch_bounds__(256) void sampler_bwd_kernel(const float* __restrict__ offsets, const float* __restrict__ anchors,
                                                          const float* __restrict__ noise_g, const float* __restrict__ unit_dir,
                                                          const float* __restrict__ gmeans, const float* __restrict__ gstds,
                                                          const int32_t* __restrict__ perm, const float* __restrict__ d_dist,
                                                          const float* __restrict__ d_z, const float* __restrict__ kl_saved,
                                                          const float* __restrict__ g_loss_kl, const float* __restrict__ g_gmeans,
                                                          const float* __restrict__ g_gstds, int R, int U, int G, int P, int N,
                                                          float base_std, float* __restrict__ d_offsets) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float m[MAXG], s[MAXG], am[MAXG], as_[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        bool ok = g < G;
        m[g] = ok ? gmeans[(size_t)r * G + g] : 0.f;
        s[g] = ok ? gstds[(size_t)r * G + g] : 1.f;
        am[g] = 0.f;
        as_[g] = 0.f;
    }
    const float uz = unit_dir[3 * r + 2];
    for (int i = lane; i < N; i += 64) {
        int o = perm[(size_t)r * N + i];
        if (o >= U) {
            int jj = o - U, g = jj / P;
            float nz = noise_g[(size_t)r * G * P + jj];
            float tot = d_dist[(size_t)r * N + i] + d_z[(size_t)r * N + i] * uz;
#pragma unroll
            for (int gg = 0; gg < MAXG; ++gg) {
                if (gg == g) {
                    float raw = m[gg] + nz * s[gg];
                    if (!(raw < 0.1f)) {  // the 0.1 clamp blocks the gradient, utils.py:214
                        am[gg] += tot;
                        as_[gg] += tot * nz;
                    }
                }
            }
        }
    }
    const float gkl = g_loss_kl ? g_loss_kl[r] : 0.f;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        am[g] = wave_sum(am[g]);
        as_[g] = wave_sum(as_[g]);
        if (g < G && lane == 0) {
            size_t q = (size_t)r * G + g;
            float m2 = kl_saved[q * 3], s2 = kl_saved[q * 3 + 1], mk = kl_saved[q * 3 + 2];
            float s1 = s[g], m1 = m[g];
            float scale = gkl * mk / (float)G;
            // d/dm1, d/ds1 of log(s2/s1 + 1e-8) + (s1^2 + (m1-m2)^2) / (2 s2^2) - 0.5
            float dm = (m1 - m2) / (s2 * s2) * scale;
            float ds = (-(s2 / (s1 * s1)) / (s2 / s1 + 1e-8f) + s1 / (s2 * s2)) * scale;
            float tm = am[g] + dm + (g_gmeans ? g_gmeans[q] : 0.f);
            float ts = as_[g] + ds + (g_gstds ? g_gstds[q] : 0.f);
            float o0 = offsets[q * 2], o1 = offsets[q * 2 + 1];
            d_offsets[q * 2] = (anchors[g] + o0 > 0.f) ? tm : 0.f;       // relu of scenerf.py:591
            d_offsets[q * 2 + 1] = (o1 + base_std > 0.f) ? ts : 0.f;    // relu of scenerf.py:593
        }
    }
}

// composite_bwd_kernel + sampler_bwd_kernel in ONE launch: the gradients w.r.t. the sorted distances and depths never leave the
// wave -- they are summed per gaussian straight from the registers through the sort permutation (utils.py:213-219 backward, the relu of
// scenerf.py:591-594, kl_gauss) -- d_dist / d_z are written only if the caller asks for them.  Same arithmetic as the two stage kernels.
template <int C>
__global__ __launch_bounds__(256) void ray_tail_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ dist,
                                                           const float* __restrict__ zv, int R, int N,
                                                           const float* __restrict__ g_depth, const float* __restrict__ g_color,
                                                           const float* __restrict__ g_weights, const float* __restrict__ g_alphas,
                                                           const float* __restrict__ g_dens, const float* __restrict__ g_zvol,
                                                           float* __restrict__ d_logits, float* __restrict__ d_dist,
                                                           float* __restrict__ d_z,
                                                           const float* __restrict__ offsets, const float* __restrict__ anchors,
                                                           const float* __restrict__ noise_g, const float* __restrict__ unit_dir,
                                                           const float* __restrict__ gmeans, const float* __restrict__ gstds,
                                                           const int32_t* __restrict__ perm, const float* __restrict__ kl_saved,
                                                           const float* __restrict__ g_loss_kl, const float* __restrict__ g_gmeans,
                                                           const float* __restrict__ g_gstds, int U, int G, int P, float base_std,
                                                           float* __restrict__ d_offsets, const float* __restrict__ g_wat,
                                                           const float* __restrict__ g_clo, const int32_t* __restrict__ closest_idx) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const size_t base = (size_t)r * N;
    float d[C], z[C], sg[C], al[C], w[C], cr[C], cg[C], cb[C], dl[C], Ti[C], o3[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        const float4 lg = ok ? *(const float4*)(logits + (base + i) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float dd = ok ? dist[base + i] : 0.f;
        d[c] = dd < 0.f ? 0.f : dd;
        z[c] = ok ? zv[base + i] : 0.f;
        o3[c] = lg.w;
        sg[c] = softplus_m1(lg.w);
        cr[c] = sigmoidf(lg.x);
        cg[c] = sigmoidf(lg.y);
        cb[c] = sigmoidf(lg.z);
    }
    float gd = g_depth[r];
    const float gcr = g_color[3 * r], gcg = g_color[3 * r + 1], gcb = g_color[3 * r + 2];
    float carryT = 1.f, carry_d = 0.f;
    float gw[C];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        float before = __shfl_up(d[c], 1, WAVE);
        if (lane == 0) before = carry_d;
        dl[c] = (i == 0) ? d[c] : d[c] - before;
        al[c] = ok ? 1.f - expf(-dl[c] * sg[c]) : 0.f;
        const float sfac = 1.f - al[c] + 1e-10f;
        const float excl = wave_excl_prod(sfac, lane);
        Ti[c] = carryT * excl;
        w[c] = al[c] * Ti[c];
        carryT *= wave_total_prod_from_excl(excl, sfac);
        carry_d = __shfl(d[c], 63, WAVE);
    }
    // weights_at_depth = weights[k] and closest_pts_to_depth = |depth - z_k| at k = argmin |depth - z| (scenerf.py:729-736; the index itself
    // carries no gradient): their gradients join the weights' at k, the rendered depth's, and z_k's
    int kci = -1;
    float s_clo = 0.f, gwat = 0.f;
    if (g_wat || g_clo) {
        kci = closest_idx[r];
        if (g_wat) gwat = g_wat[r];
        if (g_clo) {
            float sdz = 0.f, zk = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                sdz += w[c] * z[c];                                  // the rendered depth, summed as the forward sums it
                const float t = __shfl(z[c], kci & 63, WAVE);
                if ((kci >> 6) == c) zk = t;
            }
            sdz = wave_sum(sdz);
            const float df = sdz - zk;
            s_clo = g_clo[r] * (df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f));     // d|x| = sign(x), 0 at 0 like torch.abs
            gd += s_clo;
        }
    }
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        float g = gd * z[c] + gcr * cr[c] + gcg * cg[c] + gcb * cb[c];
        if (g_weights && ok) g += g_weights[base + i];
        if (i == kci) g += gwat;
        gw[c] = ok ? g : 0.f;
    }
    // the sampler's part: per-gaussian sums of (dL/ddist_i + dL/dz_i * unit_z) over the samples that came from that gaussian
    float m[MAXG], s[MAXG], am[MAXG], as_[MAXG];
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        bool ok = g < G;
        m[g] = ok ? gmeans[(size_t)r * G + g] : 0.f;
        s[g] = ok ? gstds[(size_t)r * G + g] : 1.f;
        am[g] = 0.f;
        as_[g] = 0.f;
    }
    const float uz = unit_dir[3 * r + 2];
    float tot_c[C];            // dL/ddist_i + dL/dz_i * unit_z of this lane's sample of segment c
    float carryS = 0.f;        // sum of gw*w over all later segments
    float next_first = 0.f;    // gdelta of the first sample of the next segment (i + 1 for lane 63)
#pragma unroll
    for (int c = C - 1; c >= 0; --c) {
        const int i = c * 64 + lane;
        const bool ok = i < N;
        tot_c[c] = 0.f;
        const float v = gw[c] * w[c];
        const float S = carryS + wave_excl_suffix_sum(v, lane);
        carryS += wave_sum(v);
        const float s_i = 1.f - al[c] + 1e-10f;
        float ga = gw[c] * Ti[c] - S / s_i;
        if (g_alphas && ok) ga += g_alphas[base + i];
        const float one_m = expf(-dl[c] * sg[c]);  // = 1 - alpha
        float gs = ga * dl[c] * one_m;
        if (g_dens && ok) gs += g_dens[base + i];
        const float gdelta = ok ? ga * sg[c] * one_m : 0.f;
        float after = __shfl_down(gdelta, 1, WAVE);
        if (lane == 63) after = next_first;
        next_first = __shfl(gdelta, 0, WAVE);
        if (ok) {
            const float y = o3[c] - 1.f;
            const float dsig = y > 20.f ? 1.f : sigmoidf(y);  // softplus'
            float4 o;
            o.x = gcr * w[c] * cr[c] * (1.f - cr[c]);
            o.y = gcg * w[c] * cg[c] * (1.f - cg[c]);
            o.z = gcb * w[c] * cb[c] * (1.f - cb[c]);
            o.w = gs * dsig;
            *(float4*)(d_logits + (base + i) * 4) = o;
            float gz = gd * w[c];
            if (g_zvol) gz += g_zvol[base + i];
            if (i == kci) gz -= s_clo;
            const float gdist = gdelta - ((i + 1 < N) ? after : 0.f);
            if (d_z) d_z[base + i] = gz;
            if (d_dist) d_dist[base + i] = gdist;
            tot_c[c] = gdist + gz * uz;
        }
    }
    // (segments in ascending order, like sampler_bwd_kernel's sample loop: the per-gaussian sums are then bit-identical to the stage's)
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int i = c * 64 + lane;
        if (i < N) {
            const int o_ = perm[base + i];
            if (o_ >= U) {
                const int jj = o_ - U, g = jj / P;
                const float nz = noise_g[(size_t)r * G * P + jj];
                const float tot = tot_c[c];
#pragma unroll
                for (int gg = 0; gg < MAXG; ++gg) {
                    if (gg == g) {
                        const float raw = m[gg] + nz * s[gg];
                        if (!(raw < 0.1f)) {  // the 0.1 clamp blocks the gradient, utils.py:214
                            am[gg] += tot;
                            as_[gg] += tot * nz;
                        }
                    }
                }
            }
        }
    }
    const float gkl = g_loss_kl ? g_loss_kl[r] : 0.f;
#pragma unroll
    for (int g = 0; g < MAXG; ++g) {
        am[g] = wave_sum(am[g]);
        as_[g] = wave_sum(as_[g]);
        if (g < G && lane == 0) {
            size_t q = (size_t)r * G + g;
            float m2 = kl_saved[q * 3], s2 = kl_saved[q * 3 + 1], mk = kl_saved[q * 3 + 2];
            float s1 = s[g], m1 = m[g];
            float scale = gkl * mk / (float)G;
            float dm = (m1 - m2) / (s2 * s2) * scale;
            float ds = (-(s2 / (s1 * s1)) / (s2 / s1 + 1e-8f) + s1 / (s2 * s2)) * scale;
            float tm = am[g] + dm + (g_gmeans ? g_gmeans[q] : 0.f);
            float ts = as_[g] + ds + (g_gstds ? g_gstds[q] : 0.f);
            float o0 = offsets[q * 2], o1 = offsets[q * 2 + 1];
            d_offsets[q * 2] = (anchors[g] + o0 > 0.f) ? tm : 0.f;       // relu of scenerf.py:591
            d_offsets[q * 2 + 1] = (o1 + base_std > 0.f) ? ts : 0.f;    // relu of scenerf.py:593
        }
    }
}

// ------------------------------------------------------------------------------------------------ layout changes
// in [A][B] (fp32) -> out [B][A] (TO).  RA x RB tiles through LDS, 16-byte global accesses on both sides: float4 reads along B,
// and 4 (fp32) / 8 (bf16) consecutive A-elements per store.  The extent along the CHANNEL axis is 80 when the channel count is a
// multiple of 80 (the encoder's are 80 * 2^k: with 64-wide tiles every second tile of the 80-channel map was 3/4 empty), 64
// otherwise.  Odd LDS row stride: the column reads of the write phase are conflict-free.
template <typename TO, int RA, int RB>
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ in, TO* __restrict__ out, int A, int B) {
    __shared__ float tile[RA][RB + 1];
    const int b0 = blockIdx.x * RB, a0 = blockIdx.y * RA;
    const int t = threadIdx.x;
    const bool vec_in = (B & 3) == 0;
    // read: RA rows (a) x RB/4 float4 (b)
    constexpr int QB = RB / 4;
    for (int q = t; q < RA * QB; q += 256) {
        const int ar = q / QB, bq = (q % QB) * 4;
        const int a = a0 + ar, bb = b0 + bq;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (a < A) {
            if (vec_in && bb + 3 < B) {
                const float4 f = *(const float4*)(in + (size_t)a * B + bb);
                v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (bb + e < B) v[e] = in[(size_t)a * B + bb + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[ar][bq + e] = v[e];
    }
    __syncthreads();
    // write: RB rows (b) x (RA / VN) vectors of VN consecutive a
    constexpr int VN = 16 / (int)sizeof(TO);
    constexpr int VPR = RA / VN;                // vectors per output row
    const bool vec_out = (A % VN) == 0;
    for (int q = t; q < RB * VPR; q += 256) {
        const int br = q / VPR, aq = (q % VPR) * VN;
        const int bb = b0 + br, a = a0 + aq;
        if (bb >= B) continue;
        float v[VN];
#pragma unroll
        for (int e = 0; e < VN; ++e) v[e] = tile[aq + e][br];
        if (vec_out && a + VN - 1 < A) {
            if constexpr (sizeof(TO) == 2) {
                *(uint4*)((bf16_t*)out + (size_t)bb * A + a) = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]),
                                                                            pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7]));
            } else {
                *(float4*)((float*)out + (size_t)bb * A + a) = make_float4(v[0], v[1], v[2], v[3]);
            }
        } else {
#pragma unroll
            for (int e = 0; e < VN; ++e)
                if (a + e < A) ActIO<TO>::st(out, (size_t)bb * A + a + e, v[e]);
        }
    }
}

// channels_are_a: the channel axis is A (CHW -> HWC) or B (HWC -> CHW)
template <typename TO>
static void launch_transpose(const float* in, TO* out, int A, int B, bool channels_are_a, hipStream_t s) {
    const int C = channels_are_a ? A : B;
    if (C % 80 == 0) {
        if (channels_are_a) transpose_kernel<TO, 80, 64><<<dim3(cdiv(B, 64), cdiv(A, 80)), 256, 0, s>>>(in, out, A, B);
        else transpose_kernel<TO, 64, 80><<<dim3(cdiv(B, 80), cdiv(A, 64)), 256, 0, s>>>(in, out, A, B);
    } else {
        transpose_kernel<TO, 64, 64><<<dim3(cdiv(B, 64), cdiv(A, 64)), 256, 0, s>>>(in, out, A, B);
    }
}

// ================================================================================================ C ABI
static GatherConsts make_gc(const scenerf_cfg* cfg) {
    GatherConsts gc;
    int off = 0;
    for (int s = 0; s < 5; ++s) {
        gc.C[s] = cfg->map_C[s];
        gc.Hm[s] = cfg->map_H[s];
        gc.Wm[s] = cfg->map_W[s];
        gc.Hd[s] = cfg->div_H[s];
        gc.Wd[s] = cfg->div_W[s];
        gc.off[s] = off;
        gc.chw[s] = cfg->map_chw[s];
        off += cfg->map_C[s];
    }
    return gc;
}

static int check_cfg(const scenerf_cfg* cfg) {
    SRF_CHECK(cfg != nullptr, "cfg is NULL");
    SRF_CHECK(cfg->n_gaussians >= 1 && cfg->n_gaussians <= SCENERF_MAX_GAUSSIANS, "n_gaussians %d not in [1,%d]",
              cfg->n_gaussians, SCENERF_MAX_GAUSSIANS);
    SRF_CHECK(cfg->n_pts_uni >= 0 && cfg->n_pts_per_gaussian >= 1, "bad sample counts");
    int n = (cfg->n_pts_uni > 0 ? cfg->n_pts_uni : 0) + cfg->n_gaussians * cfg->n_pts_per_gaussian;
    if (cfg->flags & SCENERF_FLAG_UNIFORM_ONLY) {   // scenerf.py:647-650: only the uniform samples are rendered
        SRF_CHECK(cfg->n_pts_uni > 0 && cfg->n_pts_per_gaussian == 1, "uniform-only: needs n_pts_uni > 0 and n_pts_per_gaussian == 1");
        n = cfg->n_pts_uni;
    }
    SRF_CHECK(cfg->n_samples == n, "n_samples %d != %d (U + G*P, or U in the uniform-only branch)", cfg->n_samples, n);
    SRF_CHECK(n <= SCENERF_MAX_SAMPLES, "n_samples %d > %d", n, SCENERF_MAX_SAMPLES);
    int csum = 0;
    for (int s = 0; s < 5; ++s) {
        SRF_CHECK(cfg->map_C[s] > 0 && cfg->map_C[s] % 16 == 0, "map_C[%d]=%d must be a positive multiple of 16", s, cfg->map_C[s]);
        SRF_CHECK(cfg->div_W[s] > 0 && cfg->div_H[s] > 0 && cfg->map_W[s] > 0 && cfg->map_H[s] > 0, "bad map dims at scale %d", s);
        csum += cfg->map_C[s];
    }
    SRF_CHECK(csum == SCENERF_D_LATENT, "sum(map_C)=%d != %d", csum, SCENERF_D_LATENT);
    SRF_CHECK(cfg->precision == 0 || cfg->precision == 1, "precision must be 0 (fp32) or 1 (bf16)");
    return 0;
}

extern "C" {

int scenerf_hip_maps_chw_to_hwc(const float* chw, void* hwc, int C, int H, int W, int precision, scenerf_stream_t stream) {
    SRF_CHECK(chw && hwc && C > 0 && H > 0 && W > 0, "maps_chw_to_hwc: bad args");
    hipStream_t s = as_stream(stream);
    int A = C, B = H * W;
    SrfLaunchScope ps(s, "maps_chw_to_hwc", 0, (double)A * B * (4 + (precision ? 2 : 4)));
    if (precision) launch_transpose<bf16_t>(chw, (bf16_t*)hwc, A, B, true, s);
    else launch_transpose<float>(chw, (float*)hwc, A, B, true, s);
    SRF_LAUNCH_CHECK("transpose_kernel");
    return 0;
}

// Zero a buffer with a bounded number of workgroups and streaming stores: the map-gradient accumulators (420 MB at KITTI) are zeroed on a
// side stream beside the forward's small kernels, and a full-speed fill through L2 slows every latency-bound kernel beside it 2-3 x
typedef unsigned int srf_u4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void fill_zero_kernel(srf_u4* __restrict__ dst, long long n16) {
    const srf_u4 z = {0u, 0u, 0u, 0u};
    // system-scope streaming stores (sc0 sc1 nt: written through, no line allocated in L2): beside the gaussian head's forward -- whose
    // weight stream lives in L2 -- the step is 15-20 us shorter than with `nt` alone (2.455 / 2.456 / 2.475 against 2.477 / 2.473 / 2.478 ms)
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long long)gridDim.x * 256)
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(dst + i), "v"(z) : "memory");
}
int scenerf_hip_fill_zero(void* dst, int64_t bytes, int workgroups, scenerf_stream_t stream) {
    SRF_CHECK(dst && bytes > 0 && bytes % 16 == 0 && ((uintptr_t)dst & 15) == 0 && workgroups > 0, "fill_zero: dst must be 16-byte aligned, bytes a multiple of 16");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "fill_zero", 0, (double)bytes);
    fill_zero_kernel<<<workgroups, 256, 0, s>>>((srf_u4*)dst, bytes / 16);
    SRF_LAUNCH_CHECK("fill_zero_kernel");
    return 0;
}

int scenerf_hip_grads_hwc_to_chw(const float* hwc, float* chw, int C, int H, int W, scenerf_stream_t stream) {
    SRF_CHECK(chw && hwc && C > 0 && H > 0 && W > 0, "grads_hwc_to_chw: bad args");
    hipStream_t s = as_stream(stream);
    int A = H * W, B = C;
    SrfLaunchScope ps(s, "grads_hwc_to_chw", 0, (double)A * B * 8);
    launch_transpose<float>(hwc, chw, A, B, false, s);
    SRF_LAUNCH_CHECK("transpose_kernel");
    return 0;
}

int scenerf_hip_ray_setup(const scenerf_cfg* cfg, const float* pixels, const float* inv_K, const float* T_s2i,
                          const float* lin_u, const float* noise_u, uint64_t* rng_state, int R, float* unit_dir, float* viewdir, float* dist_u,
                          scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(pixels && inv_K && T_s2i && unit_dir && viewdir && R > 0, "ray_setup: bad args");
    int U = cfg->n_pts_uni;
    SRF_CHECK(U == 0 || (lin_u && (noise_u || rng_state) && dist_u), "ray_setup: uniform buffers missing");
    hipStream_t s = as_stream(stream);
    int total = R * (U > 0 ? U : 1);
    SrfLaunchScope ps(s, "ray_setup", 0, (double)R * (8 + 24 + 8.0 * U));
    ray_setup_kernel<<<cdiv(total, 256), 256, 0, s>>>(pixels, inv_K, T_s2i, lin_u, noise_u, (unsigned long long*)rng_state, R, U,
                                                      cfg->uni_step, unit_dir, viewdir, dist_u);
    SRF_LAUNCH_CHECK("ray_setup_kernel");
    return 0;
}

int scenerf_hip_encode_points(const scenerf_cfg* cfg, const float* dist, int dist_ray_stride, int pts_per_ray,
                              const float* unit_dir, const float* viewdir, const float* K, const float* inv_K,
                              const float* T_s2i, int M, float* pts, int32_t* sphere_idx, float* xenc, void* x3,
                              scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(dist && unit_dir && viewdir && K && inv_K && T_s2i && sphere_idx && (xenc || x3), "encode_points: NULL argument");
    SRF_CHECK(M > 0 && pts_per_ray > 0 && M % pts_per_ray == 0, "encode_points: M=%d not a multiple of pts_per_ray=%d", M, pts_per_ray);
    hipStream_t s = as_stream(stream);
    SphereConsts sc{cfg->v_min, cfg->v_fov, cfg->h_min, cfg->h_fov, cfg->sphere_W, cfg->sphere_H};
    SrfLaunchScope ps(s, "encode_points", 0, (double)M * (4 + 8 + (xenc ? 4.0 * SCENERF_D_XENC : 0.0) + (x3 ? 6.0 * SCENERF_D_XENC : 0.0)));
    encode_points_kernel<<<cdiv(M, ENC_ROWS), 256, 0, s>>>(dist, dist_ray_stride, pts_per_ray, unit_dir, viewdir, K, inv_K, T_s2i,
                                                      sc, M, pts, sphere_idx, xenc, (bf16_t*)x3);
    SRF_LAUNCH_CHECK("encode_points_kernel");
    return 0;
}

// SphericalMapping.from_pixels (spherical_mapping.py:80-97) for a list of pixels: the encoder's image->sphere grid.  Same operation
// sequence as the renderer's per-sample index (sphere_exact.h), so the map the encoder writes and the texels the renderer reads follow one rule.
__global__ void pixels_to_sphere_kernel(const float* __restrict__ pix, const float* __restrict__ iK, srf_sphere_consts sc, int64_t M,
                                        int64_t* __restrict__ idx, float* __restrict__ dist) {
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    const float u = pix[2 * m], v = pix[2 * m + 1];
    float ox, oy;
    srf_pix_to_sphere_f(iK, sc, u, v, &ox, &oy);
    idx[2 * m] = (int64_t)srf_round_index(ox);
    idx[2 * m + 1] = (int64_t)srf_round_index(oy);
    if (dist) {
        const float cx = srf_dot3(iK[0], iK[1], iK[2], u, v, 1.0f), cy = srf_dot3(iK[3], iK[4], iK[5], u, v, 1.0f),
                    cz = srf_dot3(iK[6], iK[7], iK[8], u, v, 1.0f);
        dist[m] = srf_norm3(cx, cy, cz);
    }
}
int scenerf_hip_pixels_to_sphere(const float* pix, const float* inv_K, float v_min, float v_fov, float h_min, float h_fov, int sphere_W,
                                 int sphere_H, int64_t M, int64_t* sphere_idx, float* dist, scenerf_stream_t stream) {
    SRF_CHECK(pix && inv_K && sphere_idx && M > 0 && sphere_W > 1 && sphere_H > 1, "pixels_to_sphere: bad argument");
    hipStream_t s = as_stream(stream);
    srf_sphere_consts sc;
    sc.v_min = v_min; sc.v_fov = v_fov; sc.h_min = h_min; sc.h_fov = h_fov; sc.W = sphere_W; sc.H = sphere_H;
    SrfLaunchScope ps(s, "pixels_to_sphere", 0, (double)M * (8 + 16 + (dist ? 4 : 0)));
    pixels_to_sphere_kernel<<<(unsigned)((M + 255) / 256), 256, 0, s>>>(pix, inv_K, sc, M, sphere_idx, dist);
    SRF_LAUNCH_CHECK("pixels_to_sphere_kernel");
    return 0;
}

// test hook: the two device routines on their own (tests hold them to torch's own SLEEF build, bit for bit)
__global__ void test_acos_atan2_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, float* __restrict__ acos_a,
                                       float* __restrict__ atan2_ab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (acos_a) acos_a[i] = srf_acosf_u10(a[i]);
    if (atan2_ab) atan2_ab[i] = srf_atan2f_u10(a[i], b[i]);
}
int scenerf_hip_test_acos_atan2(const float* a, const float* b, int64_t n, float* acos_a, float* atan2_ab, scenerf_stream_t stream) {
    SRF_CHECK(a && n > 0 && (acos_a || atan2_ab) && (!atan2_ab || b), "test_acos_atan2: bad argument");
    hipStream_t s = as_stream(stream);
    test_acos_atan2_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(a, b, n, acos_a, atan2_ab);
    SRF_LAUNCH_CHECK("test_acos_atan2_kernel");
    return 0;
}

int scenerf_hip_gather_features(const scenerf_cfg* cfg, const void* const maps_hwc[SCENERF_N_SCALES],
                                const int32_t* sphere_idx, int M, void* Z, uint8_t* tile_mask, int32_t* tap_texel,
                                float* tap_weight, scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(maps_hwc && sphere_idx && Z && tile_mask && tap_texel && tap_weight && M > 0, "gather_features: NULL argument");
    MapPtrs mp;
    for (int i = 0; i < 5; ++i) {
        SRF_CHECK(maps_hwc[i] != nullptr, "gather_features: map %d is NULL", i);
        mp.p[i] = maps_hwc[i];
    }
    hipStream_t s = as_stream(stream);
    GatherConsts gc = make_gc(cfg);
    int tiles = cdiv(M, SCENERF_TILE_ROWS);
    SrfLaunchScope ps(s, "gather_features", 0, 0);
    // small launches (the gaussian head: R x G points): one block per (tile, level) instead of one per tile
    const dim3 grid(tiles, tiles < 512 ? SCENERF_N_SCALES : 1);
    if (cfg->precision)
        (tiles > 3072 ? gather_kernel<bf16_t, 1> : gather_kernel<bf16_t, 3>)<<<grid, 256, 0, s>>>(mp, gc, sphere_idx, M, (bf16_t*)Z, tile_mask, tap_texel, tap_weight);
    else
        (tiles > 3072 ? gather_kernel<float, 1> : gather_kernel<float, 3>)<<<grid, 256, 0, s>>>(mp, gc, sphere_idx, M, (float*)Z, tile_mask, tap_texel, tap_weight);
    SRF_LAUNCH_CHECK("gather_kernel");
    return 0;
}

int scenerf_hip_gaussian_sample_sort(const scenerf_cfg* cfg, const float* offsets, const float* anchors, const float* dist_u,
                                     float* noise_g, uint64_t* rng_state, const float* unit_dir, int R, float* gmeans, float* gstds,
                                     float* dist_sorted, float* z_sorted, int32_t* perm, scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(offsets && anchors && noise_g && unit_dir && gmeans && gstds && dist_sorted && z_sorted && perm && R > 0,
              "gaussian_sample_sort: NULL argument");
    SRF_CHECK(cfg->n_pts_uni == 0 || dist_u, "gaussian_sample_sort: dist_u missing");
    int N = cfg->n_samples, NP = 2;
    while (NP < N) NP <<= 1;
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "gaussian_sample_sort", 0, (double)R * N * 20);
    gaussian_sample_sort_kernel<<<R, 64, 0, s>>>(offsets, anchors, dist_u, noise_g, (unsigned long long*)rng_state, unit_dir, R, cfg->n_pts_uni,
                                                 cfg->n_gaussians, cfg->n_pts_per_gaussian, N, NP, cfg->base_std,
                                                 cfg->gauss_floor, gmeans, gstds, dist_sorted, z_sorted, perm);
    SRF_LAUNCH_CHECK("gaussian_sample_sort_kernel");
    return 0;
}

int scenerf_hip_composite_forward(const float* logits, const float* dist_sorted, const float* z_sorted, int R, int N,
                                  float* densities, float* alphas, float* weights, float* depth, float* color, float* closest,
                                  float* weights_at_depth, int32_t* closest_idx, scenerf_stream_t stream) {
    SRF_CHECK(logits && dist_sorted && z_sorted && densities && alphas && weights && depth && color && closest &&
                  weights_at_depth && closest_idx, "composite_forward: NULL argument");
    SRF_CHECK(R > 0 && N > 0 && N <= SCENERF_MAX_SAMPLES, "composite_forward: bad R=%d N=%d", R, N);
    hipStream_t s = as_stream(stream);
    dim3 grid(cdiv(R, 4));
    // algorithmic bytes: 32*N + 24 per ray
    SrfLaunchScope ps(s, "composite_fwd", 0, (double)R * (32.0 * N + 24.0));
#define CF(C) composite_fwd_kernel<C><<<grid, 256, 0, s>>>(logits, dist_sorted, z_sorted, R, N, densities, alphas, weights, depth, color, closest, weights_at_depth, closest_idx)
    if (N <= 64) CF(1);
    else if (N <= 128) CF(2);
    else if (N <= 256) CF(4);
    else CF(8);
#undef CF
    SRF_LAUNCH_CHECK("composite_fwd_kernel");
    return 0;
}

int scenerf_hip_composite_backward(const float* logits, const float* dist_sorted, const float* z_sorted, int R, int N,
                                   const float* g_depth, const float* g_color, const float* g_weights, const float* g_alphas,
                                   const float* g_densities, const float* g_zvol, float* d_logits, float* d_dist, float* d_z,
                                   scenerf_stream_t stream) {
    SRF_CHECK(logits && dist_sorted && z_sorted && g_depth && g_color && d_logits && d_dist && d_z, "composite_backward: NULL argument");
    SRF_CHECK(R > 0 && N > 0 && N <= SCENERF_MAX_SAMPLES, "composite_backward: bad R=%d N=%d", R, N);
    hipStream_t s = as_stream(stream);
    dim3 grid(cdiv(R, 4));
    SrfLaunchScope ps(s, "composite_bwd", 0, (double)R * (48.0 * N + 40.0));
#define CB(C) composite_bwd_kernel<C><<<grid, 256, 0, s>>>(logits, dist_sorted, z_sorted, R, N, g_depth, g_color, g_weights, g_alphas, g_densities, g_zvol, d_logits, d_dist, d_z)
    if (N <= 64) CB(1);
    else if (N <= 128) CB(2);
    else if (N <= 256) CB(4);
    else CB(8);
#undef CB
    SRF_LAUNCH_CHECK("composite_bwd_kernel");
    return 0;
}

int scenerf_hip_raysom_forward(const scenerf_cfg* cfg, const float* gmeans, const float* gstds, const float* dist_sorted,
                               const float* alphas, int R, float* loss_kl, float* som_means, float* som_vars, float* kl_saved,
                               uint8_t* bmu_out, scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(gmeans && gstds && dist_sorted && alphas && loss_kl && som_means && som_vars && kl_saved && R > 0,
              "raysom_forward: NULL argument");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "raysom_fwd", 0, (double)R * cfg->n_samples * 16);
    if (cfg->n_gaussians <= 4)
        raysom_fwd_kernel<4><<<cdiv(R, 4), 256, 0, s>>>(gmeans, gstds, dist_sorted, alphas, R, cfg->n_samples, cfg->n_gaussians, cfg->som_sigma,
                                                        cfg->kl_std_floor, loss_kl, som_means, som_vars, kl_saved, bmu_out);
    else
        raysom_fwd_kernel<MAXG><<<cdiv(R, 4), 256, 0, s>>>(gmeans, gstds, dist_sorted, alphas, R, cfg->n_samples, cfg->n_gaussians, cfg->som_sigma,
                                                           cfg->kl_std_floor, loss_kl, som_means, som_vars, kl_saved, bmu_out);
    SRF_LAUNCH_CHECK("raysom_fwd_kernel");
    return 0;
}

int scenerf_hip_sampler_backward(const scenerf_cfg* cfg, const float* offsets, const float* anchors, const float* noise_g,
                                 const float* unit_dir, const float* gmeans, const float* gstds, const int32_t* perm,
                                 const float* d_dist, const float* d_z, const float* kl_saved, const float* g_loss_kl,
                                 const float* g_gmeans, const float* g_gstds, int R, float* d_offsets, scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(offsets && anchors && noise_g && unit_dir && gmeans && gstds && perm && d_dist && d_z && kl_saved && d_offsets && R > 0,
              "sampler_backward: NULL argument");
    hipStream_t s = as_stream(stream);
    SrfLaunchScope ps(s, "sampler_bwd", 0, (double)R * cfg->n_samples * 16);
    sampler_bwd_kernel<<<cdiv(R, 4), 256, 0, s>>>(offsets, anchors, noise_g, unit_dir, gmeans, gstds, perm, d_dist, d_z, kl_saved,
                                                  g_loss_kl, g_gmeans, g_gstds, R, cfg->n_pts_uni, cfg->n_gaussians,
                                                  cfg->n_pts_per_gaussian, cfg->n_samples, cfg->base_std, d_offsets);
    SRF_LAUNCH_CHECK("sampler_bwd_kernel");
    return 0;
}

int scenerf_hip_ray_tail_forward(const scenerf_cfg* cfg, const float* logits, const float* dist_sorted, const float* z_sorted,
                                 const float* gmeans, const float* gstds, int R, float* densities, float* alphas, float* weights,
                                 float* depth, float* color, float* closest, float* weights_at_depth, int32_t* closest_idx,
                                 float* loss_kl, float* som_means, float* som_vars, float* kl_saved, uint8_t* bmu_out,
                                 scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(logits && dist_sorted && z_sorted && depth && color && closest && weights_at_depth && closest_idx && R > 0, "ray_tail_forward: NULL argument");
    const bool som = loss_kl != nullptr;      // the RaySOM half: all of its outputs or none
    SRF_CHECK(som ? (gmeans && gstds && som_means && som_vars && kl_saved && alphas) : (!som_means && !som_vars && !kl_saved && !bmu_out),
              "ray_tail_forward: loss_kl, som_means, som_vars, kl_saved (and gmeans, gstds, alphas) go together");
    const int N = cfg->n_samples;
    hipStream_t s = as_stream(stream);
    dim3 grid(cdiv(R, 4));
    SrfLaunchScope ps(s, som ? "ray_tail_fwd" : "ray_tail_fwd_nosom", 0,
                      (double)R * ((20.0 + (densities ? 4.0 : 0.0) + (alphas ? 4.0 : 0.0) + (weights ? 4.0 : 0.0)) * N + 24.0));
#define TF3(C, GB, SM) ray_tail_fwd_kernel<C, GB, SM><<<grid, 256, 0, s>>>(logits, dist_sorted, z_sorted, gmeans, gstds, R, N, cfg->n_gaussians, cfg->som_sigma, cfg->kl_std_floor, densities, alphas, weights, depth, color, closest, weights_at_depth, closest_idx, loss_kl, som_means, som_vars, kl_saved, bmu_out)
#define TF(C) { if (!som) TF3(C, 4, false); else if (cfg->n_gaussians <= 4) TF3(C, 4, true); else TF3(C, MAXG, true); }
    if (N <= 64) TF(1)
    else if (N <= 128) TF(2)
    else if (N <= 256) TF(4)
    else TF(8)
#undef TF
#undef TF3
    SRF_LAUNCH_CHECK("ray_tail_fwd_kernel");
    return 0;
}

int scenerf_hip_ray_tail_backward(const scenerf_cfg* cfg, const float* logits, const float* dist_sorted, const float* z_sorted, int R,
                                  const float* g_depth, const float* g_color, const float* g_weights, const float* g_alphas,
                                  const float* g_densities, const float* g_zvol, const float* offsets, const float* anchors,
                                  const float* noise_g, const float* unit_dir, const float* gmeans, const float* gstds,
                                  const int32_t* perm, const float* kl_saved, const float* g_loss_kl, const float* g_gmeans,
                                  const float* g_gstds, float* d_logits, float* d_offsets, float* d_dist, float* d_z,
                                  const float* g_weights_at_depth, const float* g_closest, const int32_t* closest_idx,
                                  scenerf_stream_t stream) {
    if (check_cfg(cfg)) return 1;
    SRF_CHECK(logits && dist_sorted && z_sorted && g_depth && g_color && d_logits && offsets && anchors && noise_g && unit_dir && gmeans &&
                  gstds && perm && kl_saved && d_offsets && R > 0, "ray_tail_backward: NULL argument");
    const int N = cfg->n_samples;
    hipStream_t s = as_stream(stream);
    dim3 grid(cdiv(R, 4));
    SrfLaunchScope ps(s, "ray_tail_bwd", 0, (double)R * (44.0 * N + 40.0));
#define TB(C) ray_tail_bwd_kernel<C><<<grid, 256, 0, s>>>(logits, dist_sorted, z_sorted, R, N, g_depth, g_color, g_weights, g_alphas, g_densities, g_zvol, d_logits, d_dist, d_z, offsets, anchors, noise_g, unit_dir, gmeans, gstds, perm, kl_saved, g_loss_kl, g_gmeans, g_gstds, cfg->n_pts_uni, cfg->n_gaussians, cfg->n_pts_per_gaussian, cfg->base_std, d_offsets, g_weights_at_depth, g_closest, closest_idx)
    SRF_CHECK(!(g_weights_at_depth || g_closest) || closest_idx, "ray_tail_backward: gradients of weights_at_depth / closest_pts_to_depth need closest_idx");
    if (N <= 64) TB(1);
    else if (N <= 128) TB(2);
    else if (N <= 256) TB(4);
    else TB(8);
#undef TB
    SRF_LAUNCH_CHECK("ray_tail_bwd_kernel");
    return 0;
}

}  // extern "C"
