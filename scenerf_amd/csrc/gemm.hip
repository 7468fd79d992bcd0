// MFMA GEMM kernels for the ResnetFC pass, written for gfx950 (wave64, v_mfma_f32_32x32x16_bf16 /
// v_mfma_f32_32x32x2_f32).  Two kernels:
//   gemm_nt : C[M][N] = epi([A1 | Z-segments] @ W^T)      forward layers, dgrad, feature-gradient scatter
//   gemm_tn : C[N][K] += D^T @ act(A)  (+ column sums of D) weight and bias gradients (contraction over the M rows)
//
// gemm_nt comes in two tile shapes:
//   wide  128 x 512, 8 waves (2 x 4, each 64 x 128 = 2 x 4 MFMA 32x32 blocks, 128 accumulator VGPRs).  A CU can
//         only pull ~10 B/clk from HBM, so the hidden-layer GEMMs (N = 512) let ONE workgroup own all 512 output
//         columns of its 128 rows: the activation stream is read from HBM exactly once and the 512 x K weight panel
//         streams from L2 (shared by all 1200 row tiles).
//   small 128 x 128, 4 waves (2 x 2, each 64 x 64) for narrow outputs (per-scale feature gradients, N = 80..1280),
//         small M (the gaussian head: 4 points per ray) and the unit tests.
// K is streamed in chunks through a double-buffered LDS image whose rows are padded by 16 bytes (conflict-free
// ds_read_b128 fragment reads); the next chunk is prefetched global->VGPR while the MFMAs of the current one run.
// Both element types use the same fragment indexing: lane l holds k = (l>>5)*8 .. +8 of row (l&31); for fp32 the
// 8 values feed 8 back-to-back 32x32x2 MFMAs (any bijective k-order is valid as long as A and B agree).
#include <stdlib.h>

#include <vector>

#include "gemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define BM 128
#define MAX_CHUNKS 192
#define CLD 132  // fp32 C tile row stride in LDS (128 + 4 pad): 128*132*4 = 67584 B

template <int TJ_, int NWN_, int RB_, int PF_, int OCC_> struct TileCfg {
    static constexpr int PF = PF_;              // prefetch depth: chunks in flight global->VGPR
    static constexpr int OCC = OCC_;            // waves per SIMD the register budget must allow
    static constexpr int TJ = TJ_;              // 32-wide MFMA blocks per wave along N
    static constexpr int NWN = NWN_;            // waves along N (2 along M)
    static constexpr int RB = RB_;              // LDS row bytes: data + 16 pad
    static constexpr int BN = NWN_ * TJ_ * 32;
    static constexpr int NT = 2 * NWN_ * 64;    // threads
    static constexpr int TILE_A = BM * RB_;
    static constexpr int TILE_W = BN * RB_;
    static constexpr int STAGE = TILE_A + TILE_W;
    static constexpr int PPR = (RB_ - 16) / 16; // 16-byte pieces per full row
    static constexpr int NPA = (BM * PPR + NT - 1) / NT;
    static constexpr int NPW = (BN * PPR + NT - 1) / NT;
    static constexpr int LDS_LOOP = 2 * STAGE + MAX_CHUNKS * 16 + 16;  // operand ring + chunk table
    static constexpr int LDS_EPI = BM * CLD * 4 + BM * 4 * 8;          // fp32 C staging of one 128-column pass + taps
    static constexpr int LDS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
};
typedef TileCfg<2, 2, 144, 2, 2> CfgS;  // 128 x 128, 256 threads, 128 B of K per row per chunk
typedef TileCfg<4, 4, 80, 3, 2> CfgW;
typedef TileCfg<2, 4, 80, 2, 4> CfgM;   // 128 x 256, 512 threads (each wave 64 x 64), 2 workgroups per CU = 4 waves/SIMD  // 128 x 512, 512 threads,  64 B of K per row per chunk

__device__ static inline uint4 zero4() { return make_uint4(0u, 0u, 0u, 0u); }
__device__ static inline int pieces_shift(int ppr) { return ppr >= 8 ? 3 : ppr >= 4 ? 2 : ppr >= 2 ? 1 : 0; }
// next chunk length (elements): at most BK, a multiple of 16 elements, and a power-of-two number of 16-byte pieces
template <int ES> __device__ static inline int chunk_len(int remaining, int bk) {
    int kc = remaining < bk ? remaining : bk;
    int pieces = kc * ES / 16;
    int p2 = 1 << pieces_shift(pieces);
    int len = p2 * 16 / ES;
    return len < 16 ? 16 : len;
}

// XCD-aware block order: the dispatcher places block b on XCD b % 8 (observed, used for speed only).  Give each
// XCD one contiguous range of the linear tile index, so tiles that share an operand panel run back-to-back on the
// SAME XCD and hit its private L2.  Bijective for any block count (cdna_hip_programming.md T1).
__device__ static inline int xcd_remap(int id, int total) {
    const int q = total >> 3, r = total & 7;
    const int xcd = id & 7, slot = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

template <typename T> __device__ static inline uint4 relu16B(uint4 v);
template <> __device__ inline uint4 relu16B<bf16_t>(uint4 v) {
    return make_uint4(relu_bf16x2(v.x), relu_bf16x2(v.y), relu_bf16x2(v.z), relu_bf16x2(v.w));
}
template <> __device__ inline uint4 relu16B<float>(uint4 v) {
    // negative floats have the sign bit set (-0 -> 0 is fine)
    return make_uint4((v.x >> 31) ? 0u : v.x, (v.y >> 31) ? 0u : v.y, (v.z >> 31) ? 0u : v.z, (v.w >> 31) ? 0u : v.w);
}

// one 16-element k-step on a (64 x TJ*32) wave tile
template <typename T, int TJ, int RB> struct WaveMma;
template <int TJ, int RB> struct WaveMma<bf16_t, TJ, RB> {
    __device__ static inline void step(f32x16_t (&acc)[2][TJ], const char* As, const char* Ws, int kk, int lane, int wm, int wn,
                                       bool relu_b = false) {
        const int koff = (kk * 16 + (lane >> 5) * 8) * 2;
        uint4 a[2], b[TJ];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = *(const uint4*)(As + (wm * 64 + i * 32 + (lane & 31)) * RB + koff);
#pragma unroll
        for (int j = 0; j < TJ; ++j) b[j] = *(const uint4*)(Ws + (wn * TJ * 32 + j * 32 + (lane & 31)) * RB + koff);
        if (relu_b) {
#pragma unroll
            for (int j = 0; j < TJ; ++j) b[j] = relu16B<bf16_t>(b[j]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TJ; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]),
                                                                    __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
    }
};
template <int TJ, int RB> struct WaveMma<float, TJ, RB> {
    __device__ static inline void step(f32x16_t (&acc)[2][TJ], const char* As, const char* Ws, int kk, int lane, int wm, int wn,
                                       bool = false) {
        const int koff = (kk * 16 + (lane >> 5) * 8) * 4;
        float a[2][8], b[TJ][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* p = As + (wm * 64 + i * 32 + (lane & 31)) * RB + koff;
            float4 lo = *(const float4*)p, hi = *(const float4*)(p + 16);
            a[i][0] = lo.x; a[i][1] = lo.y; a[i][2] = lo.z; a[i][3] = lo.w;
            a[i][4] = hi.x; a[i][5] = hi.y; a[i][6] = hi.z; a[i][7] = hi.w;
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const char* p = Ws + (wn * TJ * 32 + j * 32 + (lane & 31)) * RB + koff;
            float4 lo = *(const float4*)p, hi = *(const float4*)(p + 16);
            b[j][0] = lo.x; b[j][1] = lo.y; b[j][2] = lo.z; b[j][3] = lo.w;
            b[j][4] = hi.x; b[j][5] = hi.y; b[j][6] = hi.z; b[j][7] = hi.w;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
    }
};

// epilogue of 8 consecutive columns of one row: v = acc + bias; v += res; v = mask > 0 ? v : 0; v += res2; store
template <typename T>
__device__ __forceinline__ void epi_item(const GemmNT& p, float* v, int m, int n) {
    if (p.bias) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += p.bias[n + e];
    }
    if (p.res) {
        float r[8];
        if (p.res_f32) load8<float>(p.res, (size_t)m * p.ldres + n, r);
        else load8<T>(p.res, (size_t)m * p.ldres + n, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (p.maskp) {
        float r[8];
        load8<T>(p.maskp, (size_t)m * p.ldmask + n, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = r[e] > 0.f ? v[e] : 0.f;
    }
    if (p.res2) {
        float r[8];
        load8<T>(p.res2, (size_t)m * p.ldres2 + n, r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += r[e];
    }
    if (p.out_f32) store8<float>(p.out, (size_t)m * p.ldout + n, v);
    else store8<T>(p.out, (size_t)m * p.ldout + n, v);
}

// ---- epilogue shared by the register-staged and the direct-to-LDS kernels ---------------------------------------
struct NtScatter {   // where a tile's scatter epilogue goes (the launch's single scale, or the tile's scale of a multi-scale launch)
    float* gmap; size_t st; long sc; int scale; int N;
};
template <typename T, typename CF>
__device__ __forceinline__ void nt_epilogue(const GemmNT& p, f32x16_t (&acc)[2][CF::TJ], char* lds, int m0, int n0, int tid, const NtScatter sct) {
    constexpr int TJ = CF::TJ, NWN = CF::NWN, BN = CF::BN, NT = CF::NT;
    const int lane = tid & 63, wv = tid >> 6, wm = wv / NWN, wn = wv % NWN;
    // ---- epilogue: stage 128 x 128 fp32 column passes in LDS (reusing the operand buffers), then every thread
    // handles 8 consecutive columns of a row with 16/32-byte global accesses (the MFMA C layout gives a lane one
    // column of 16 scattered rows: storing from it directly means 2-byte strided accesses) -------------------------
    float* Cs = (float*)lds;
    constexpr int NPASS = BN / 128;
    constexpr int WCOLS = TJ * 32;  // columns owned by one wave
#define CS_W(i, j, r) Cs[(wm * 64 + (i) * 32 + ((r) & 3) + 8 * ((r) >> 2) + 4 * (lane >> 5)) * CLD + cbase + (j) * 32 + (lane & 31)] = acc[i][j][r];
#define CS_TILE(i, j)                                                                                               \
    CS_W(i, j, 0) CS_W(i, j, 1) CS_W(i, j, 2) CS_W(i, j, 3) CS_W(i, j, 4) CS_W(i, j, 5) CS_W(i, j, 6) CS_W(i, j, 7) \
    CS_W(i, j, 8) CS_W(i, j, 9) CS_W(i, j, 10) CS_W(i, j, 11) CS_W(i, j, 12) CS_W(i, j, 13) CS_W(i, j, 14) CS_W(i, j, 15)
    int* s_tx = (int*)(lds + BM * CLD * 4);  // taps of this row tile (scatter mode): [128][4] texel, [128][4] weight
    float* s_tw = (float*)(s_tx + BM * 4);
    if (p.scatter_scale >= 0) {
        for (int i = tid; i < BM * 4; i += NT) {
            const int m = m0 + (i >> 2);
            const size_t o = ((size_t)m * 5 + sct.scale) * 4 + (i & 3);
            s_tx[i] = m < p.M ? p.tap_texel[o] : -1;
            s_tw[i] = m < p.M ? p.tap_weight[o] : 0.f;
        }
    }
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if ((wn * WCOLS) / 128 == pass) {
            const int cbase = (wn * WCOLS) % 128;
            CS_TILE(0, 0)
            CS_TILE(0, 1)
            CS_TILE(1, 0)
            CS_TILE(1, 1)
            if constexpr (TJ == 4) {
                CS_TILE(0, 2)
                CS_TILE(0, 3)
                CS_TILE(1, 2)
                CS_TILE(1, 3)
            }
        }
        __syncthreads();
        const int nb = n0 + pass * 128;
        if (p.scatter_scale >= 0) {
            // grid_sampler backward.  Lanes span 64 consecutive channels of one row, so each atomic instruction adds
            // a contiguous 256-byte run of the (H,W,C) gradient map.  The taps of a row depend only on its spherical
            // pixel; consecutive samples of a ray often share it (far samples converge), and same-address atomics
            // serialise in L2 -- so rows of a wave's contiguous block with identical taps are summed first and
            // scattered once.
            constexpr int RPW = BM / (NT / 64);
            const size_t gst = sct.st;
            int r = wv * RPW;
            const int rend = min(r + RPW, p.M - m0);
            while (r < rend) {
                const int t0 = s_tx[r * 4], t1 = s_tx[r * 4 + 1], t2 = s_tx[r * 4 + 2], t3 = s_tx[r * 4 + 3];
                int e = r + 1;
                while (e < rend && s_tx[e * 4] == t0 && s_tx[e * 4 + 1] == t1 && s_tx[e * 4 + 2] == t2 && s_tx[e * 4 + 3] == t3) ++e;
                if ((t0 & t1 & t2 & t3) >= 0 || t0 >= 0 || t1 >= 0 || t2 >= 0 || t3 >= 0) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int col = h * 64 + lane, n = nb + col;
                        if (n < sct.N) {
                            float v = 0.f;
                            for (int q = r; q < e; ++q) v += Cs[q * CLD + col];
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                const int tx = s_tx[r * 4 + t];
                                if (tx >= 0) unsafeAtomicAdd(sct.gmap + (size_t)tx * gst + (size_t)n * sct.sc, v * s_tw[r * 4 + t]);
                            }
                        }
                    }
                }
                r = e;
            }
        } else {
            for (int it = tid; it < BM * 16; it += NT) {
                const int row = it >> 4, cg = it & 15;
                const int m = m0 + row, n = nb + cg * 8;
                if (m < p.M && n < p.N) {
                    float v[8];
                    const float4 lo = *(const float4*)(Cs + row * CLD + cg * 8), hi = *(const float4*)(Cs + row * CLD + cg * 8 + 4);
                    v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
                    epi_item<T>(p, v, m, n);
                }
            }
        }
        if (pass + 1 < NPASS) __syncthreads();
    }
#undef CS_TILE
#undef CS_W
}

// ================================================================================================ NT
template <typename T, typename CF>
__global__ __launch_bounds__(CF::NT, CF::OCC) void gemm_nt_kernel(GemmNT p) {
    // all LDS lives in the one dynamic region: a static __shared__ would shift its base off 16-byte alignment
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int4* chunks = (int4*)(lds + 2 * CF::STAGE);  // {src, a_col, w_col, kc}
    int* s_n_ptr = (int*)(lds + 2 * CF::STAGE + MAX_CHUNKS * 16);
    constexpr int ES = (int)sizeof(T);
    constexpr int BK = (CF::RB - 16) / ES;
    constexpr int TJ = CF::TJ, NWN = CF::NWN, RB = CF::RB, BN = CF::BN, NT = CF::NT;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv / NWN, wn = wv % NWN;
    const int tiles_n = p.ms_n > 0 ? p.ms_t0[GEMM_MAX_SEG] : (p.N + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lin / tiles_n) * BM;
    int n0 = (lin % tiles_n) * BN;
    // this tile's W rows / column count / scatter target: the launch's, or its scale's (multi-scale scatter, BN == 128)
    const void* Wl = p.W;
    int Nl = p.N, skip_bit = p.skip_bit;
    NtScatter sct = {p.gmap, p.gmap_st ? (size_t)p.gmap_st : (size_t)p.N, p.gmap_sc, p.scatter_scale, p.N};
    if (p.ms_n > 0) {
        const int tn = lin % tiles_n;
        int sc = 0;
#pragma unroll
        for (int i = 1; i < GEMM_MAX_SEG; ++i)
            if (i < p.ms_n && tn >= p.ms_t0[i]) sc = i;
        n0 = (tn - p.ms_t0[sc]) * BN;
        Wl = p.ms_W[sc];
        Nl = p.ms_C[sc];
        skip_bit = sc;
        sct.gmap = p.ms_gmap[sc];
        sct.st = p.ms_st[sc] ? (size_t)p.ms_st[sc] : (size_t)Nl;
        sct.sc = p.ms_sc[sc];
        sct.scale = sc;
        sct.N = Nl;
        if (!sct.gmap) return;
    }

    unsigned mask = 0xffffffffu;
    if (p.tile_mask) mask = p.tile_mask[m0 / SCENERF_TILE_ROWS];
    if (skip_bit >= 0 && !((mask >> skip_bit) & 1u)) return;  // uniform: whole tile contributes exact zeros

    if (tid == 0) {
        int n = 0;
        for (int k0 = 0; k0 < p.K1;) {
            const int kc = chunk_len<ES>(p.K1 - k0, BK);
            chunks[n++] = make_int4(0, k0, k0, kc | (pieces_shift(kc * ES / 16) << 16));
            k0 += kc;
        }
        int wbase = p.K1;
#pragma unroll
        for (int s = 0; s < GEMM_MAX_SEG; ++s) {  // constant trip count: keeps the by-value params out of scratch
            if (s < p.nseg && ((mask >> s) & 1u)) {
                for (int k0 = 0; k0 < p.seg_len[s];) {
                    const int kc = chunk_len<ES>(p.seg_len[s] - k0, BK);
                    chunks[n++] = make_int4(1, p.seg_off[s] + k0, wbase + k0, kc | (pieces_shift(kc * ES / 16) << 16));
                    k0 += kc;
                }
            }
            if (s < p.nseg) wbase += p.seg_len[s];
        }
        *s_n_ptr = n;
    }
    __syncthreads();
    const int nch = *s_n_ptr;

    f32x16_t acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Prefetch discipline: load_chunk is branch-free straight-line code that only ISSUES global loads (indices
    // clamped, nothing consumes the data), so all loads of a chunk go out back-to-back and hipcc's s_waitcnt lands in
    // store_chunk, after the MFMAs of the current chunk.  (A guarded load sits in its own basic block and hipcc then
    // waits vmcnt(0) in front of every one of them: each load paid a full memory latency.)  ReLU and the zeroing of
    // out-of-range rows happen on the way into LDS.  Pieces per row is a power of two: row/piece via shifts.
    uint4 ra[CF::PF][CF::NPA], rw[CF::PF][CF::NPW];
    auto load_chunk = [&](int c, uint4 (&ra)[CF::NPA], uint4 (&rw)[CF::NPW]) {
        const int4 ch = chunks[c];
        const int sh = ch.w >> 16;             // log2(16-byte pieces per row)
        const int pm = (1 << sh) - 1;
        const char* Ab = (const char*)(ch.x == 0 ? p.A1 : p.A2);
        const size_t lda = (size_t)(ch.x == 0 ? p.lda1 : p.lda2) * ES;
#pragma unroll
        for (int i = 0; i < CF::NPA; ++i) {
            const int q = min(tid + i * NT, (BM << sh) - 1);
            const int gm = min(m0 + (q >> sh), p.M - 1);
            ra[i] = *(const uint4*)(Ab + (size_t)gm * lda + (size_t)ch.y * ES + (q & pm) * 16);
        }
#pragma unroll
        for (int i = 0; i < CF::NPW; ++i) {
            const int q = min(tid + i * NT, (BN << sh) - 1);
            const int gn = min(n0 + (q >> sh), Nl - 1);
            rw[i] = *(const uint4*)((const char*)Wl + ((size_t)gn * p.ldw + ch.z) * ES + (q & pm) * 16);
        }
    };
    auto store_chunk = [&](int c, int buf, const uint4 (&ra)[CF::NPA], const uint4 (&rw)[CF::NPW]) {
        const int4 ch = chunks[c];
        const int sh = ch.w >> 16;
        const int pm = (1 << sh) - 1;
        const bool relu = ch.x == 0 && p.relu1;
        char* As = lds + buf * CF::STAGE;
        char* Ws = As + CF::TILE_A;
#pragma unroll
        for (int i = 0; i < CF::NPA; ++i) {
            const int q = tid + i * NT;
            if (q < (BM << sh)) {
                const int row = q >> sh;
                uint4 v = (m0 + row < p.M) ? ra[i] : zero4();
                if (relu) v = relu16B<T>(v);
                *(uint4*)(As + row * RB + (q & pm) * 16) = v;
            }
        }
#pragma unroll
        for (int i = 0; i < CF::NPW; ++i) {
            const int q = tid + i * NT;
            if (q < (BN << sh)) {
                const int row = q >> sh;
                *(uint4*)(Ws + row * RB + (q & pm) * 16) = (n0 + row < Nl) ? rw[i] : zero4();
            }
        }
    };

    // Software pipeline, depth PF: chunk k lives in register set k % PF.  Iteration c issues the loads of chunk
    // c + PF (into the set chunk c just vacated), runs the MFMAs of chunk c from LDS buffer c & 1, then moves chunk
    // c + 1 (loaded PF - 1 iterations ago) into the other LDS buffer: the wait there is a counted vmcnt that leaves
    // the younger PF - 1 chunks in flight.  Unrolled by PF so every register-set index is a compile-time constant.
    constexpr int PF = CF::PF;
    if (nch > 0) {
#pragma unroll
        for (int u = 0; u < PF; ++u) load_chunk(u < nch ? u : nch - 1, ra[u], rw[u]);
        store_chunk(0, 0, ra[0], rw[0]);
    }
    __syncthreads();
    for (int c0 = 0; c0 < nch; c0 += PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            const int c = c0 + u;
            if (c < nch) {
                load_chunk(c + PF < nch ? c + PF : nch - 1, ra[u], rw[u]);
                const char* As = lds + (c & 1) * CF::STAGE;
                const char* Ws = As + CF::TILE_A;
                const int ks = (chunks[c].w & 0xffff) / 16;
                for (int kk = 0; kk < ks; ++kk) WaveMma<T, TJ, RB>::step(acc, As, Ws, kk, lane, wm, wn);
                if (c + 1 < nch) store_chunk(c + 1, (c + 1) & 1, ra[(u + 1) % PF], rw[(u + 1) % PF]);
                __syncthreads();
            }
        }
    }

    nt_epilogue<T, CF>(p, acc, lds, m0, n0, tid, sct);
}

// ================================================================================================ NT, direct-to-LDS
// bf16 hidden-layer variant of gemm_nt (tile 128 x 256, 8 waves, two workgroups per CU) whose operand tiles go
// HBM/L2 -> LDS with global_load_lds_dwordx4 (no VGPR staging, no ds_write pass: the ds_write path moves only
// ~80 B/clk/CU and was the busiest LDS resource of the register-staged kernel).
//   * LDS image of a stage: 384 rows (128 of A, then 256 of W) x 64 B (32 bf16 of K), unpadded, because a wave's
//     glds writes 64 lanes x 16 B = 16 whole rows contiguously.  Bank conflicts are removed by an XOR swizzle of the
//     16-byte slot inside a row, slot' = slot ^ ((row >> 2) & 3), applied on the per-lane GLOBAL source address
//     (the LDS destination of a glds is always lane-linear) and again on the fragment read.
//   * 3 stages, prefetch distance 2, ONE raw s_barrier per chunk, counted s_waitcnt vmcnt(3): the three loads a
//     wave issued for chunk c+1 stay in flight across the barrier while chunk c is consumed.
//   * ReLU on the A operand is applied to the fragment registers (8 VALU per k-step).
#define G_ROWB 64
#define G_STAGE ((BM + 256) * G_ROWB)      // 24576 B
#define G_NSTAGE 3
struct CfgG {
    static constexpr int TJ = 2, NWN = 4, BN = 256, NT = 512, OCC = 4;
    static constexpr int LDS_LOOP = G_NSTAGE * G_STAGE + MAX_CHUNKS * 16 + 16;
    static constexpr int LDS_EPI = BM * CLD * 4 + BM * 4 * 8;
    static constexpr int LDS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
};

// One 1-KiB LDS-DMA piece: lane l's 16 bytes at *g land at LDS byte (lds_wave_base + 16 l).  Issued from inline asm on
// purpose: through the builtin hipcc models the LDS write and puts s_waitcnt vmcnt(0) in front of every later ds_read,
// which serialises the pipeline; an asm load is invisible to its counters, so the ONLY waits are the counted
// s_waitcnt vmcnt(N) written in the loop below (cdna_hip_programming.md §5.7).  M0 (the DMA's LDS base) is saved and
// restored inside the same statement; the s_nop covers the M0-write -> DMA-read hazard.
__device__ static inline void glds16(const void* g, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_wave_base)
                 : "memory");
}

__global__ __launch_bounds__(512, 4) void gemm_nt_glds_kernel(GemmNT p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    typedef bf16_t T;
    constexpr int ES = 2, BK = 32, TJ = 2, NWN = 4, BN = 256;
    int4* chunks = (int4*)(lds + G_NSTAGE * G_STAGE);  // {src, a_col, w_col, kc | shift << 16}
    int* s_n_ptr = (int*)(lds + G_NSTAGE * G_STAGE + MAX_CHUNKS * 16);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv / NWN, wn = wv % NWN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (lin / tiles_n) * BM, n0 = (lin % tiles_n) * BN;

    unsigned mask = 0xffffffffu;
    if (p.tile_mask) mask = p.tile_mask[m0 / SCENERF_TILE_ROWS];
    if (p.skip_bit >= 0 && !((mask >> p.skip_bit) & 1u)) return;

    if (tid == 0) {
        int n = 0;
        for (int k0 = 0; k0 < p.K1;) {
            const int kc = chunk_len<ES>(p.K1 - k0, BK);
            chunks[n++] = make_int4(0, k0, k0, kc | (pieces_shift(kc * ES / 16) << 16));
            k0 += kc;
        }
        int wbase = p.K1;
#pragma unroll
        for (int s = 0; s < GEMM_MAX_SEG; ++s) {
            if (s < p.nseg && ((mask >> s) & 1u)) {
                for (int k0 = 0; k0 < p.seg_len[s];) {
                    const int kc = chunk_len<ES>(p.seg_len[s] - k0, BK);
                    chunks[n++] = make_int4(1, p.seg_off[s] + k0, wbase + k0, kc | (pieces_shift(kc * ES / 16) << 16));
                    k0 += kc;
                }
            }
            if (s < p.nseg) wbase += p.seg_len[s];
        }
        *s_n_ptr = n;
    }
    __syncthreads();
    const int nch = *s_n_ptr;

    f32x16_t acc[2][TJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- per-lane constants, computed once: the loop body must stay lean -- a first version recomputed row / swizzle
    // / 64-bit address math per chunk and measured 16 VALU + 9 SALU instructions per MFMA (issue-bound at 27 % MFMA).
    // LDS fragment offsets (swizzled) for the two k-steps of a 64-byte row chunk:
    int offA[2][2], offB[TJ][2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int slot = kk * 2 + (lane >> 5);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = wm * 64 + i * 32 + (lane & 31);
            offA[i][kk] = r * G_ROWB + ((slot ^ ((r >> 2) & 3)) << 4);
        }
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const int r = wn * TJ * 32 + j * 32 + (lane & 31);
            offB[j][kk] = BM * G_ROWB + r * G_ROWB + ((slot ^ ((r >> 2) & 3)) << 4);
        }
    }
    // This wave's three 1-KiB LDS-DMA pieces of a stage: piece i = wv + 8u covers image rows [16 i, 16 i + 16); u = 0
    // is always an A piece (rows 0..127), u = 1, 2 are W pieces.  Lane -> (row 16 i + lane/4, physical slot lane & 3);
    // it fetches logical slot = physical ^ ((row >> 2) & 3).  Row pointers are chunk-invariant; a chunk only adds its
    // (wave-uniform) column offset.
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const char *pA1, *pA2, *pW[2];
    int lsl[3];
    {
        const int R = 16 * wv + (lane >> 2);
        lsl[0] = (lane & 3) ^ ((R >> 2) & 3);
        const int gm = min(m0 + R, p.M - 1);
        pA1 = (const char*)p.A1 + (size_t)gm * p.lda1 * ES;
        pA2 = (const char*)p.A2 + (size_t)gm * p.lda2 * ES;
#pragma unroll
        for (int u = 1; u < 3; ++u) {
            const int Rw = 16 * (wv + 8 * u) + (lane >> 2);
            lsl[u] = (lane & 3) ^ ((Rw >> 2) & 3);
            const int gn = min(n0 + Rw - BM, p.N - 1);
            pW[u - 1] = (const char*)p.W + (size_t)gn * p.ldw * ES;
        }
    }
    auto issue = [&](int c, const int stage) {
        const int4 ch = chunks[c];
        const int ppr = 1 << (ch.w >> 16);
        const char* ga = (ch.x == 0 ? pA1 : pA2) + (size_t)ch.y * ES;
        const size_t wcol = (size_t)ch.z * ES;
        const unsigned sb = lds0 + stage * G_STAGE + wvu * 1024;
        // tail chunks (< 4 pieces per row) keep every lane inside the chunk's columns (duplicates land in unused slots)
        glds16(ga + (min(lsl[0], ppr - 1) << 4), __builtin_amdgcn_readfirstlane(sb));
        glds16(pW[0] + wcol + (min(lsl[1], ppr - 1) << 4), __builtin_amdgcn_readfirstlane(sb + 8 * 1024));
        glds16(pW[1] + wcol + (min(lsl[2], ppr - 1) << 4), __builtin_amdgcn_readfirstlane(sb + 16 * 1024));
    };
    auto compute = [&](int c, const int stage) {
        const char* S = lds + stage * G_STAGE;
        const int ks = (chunks[c].w & 0xffff) >> 4;
        const bool relu = chunks[c].x == 0 && p.relu1;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            if (kk < ks) {
                uint4 a[2], b[TJ];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = *(const uint4*)(S + offA[i][kk]);
#pragma unroll
                for (int j = 0; j < TJ; ++j) b[j] = *(const uint4*)(S + offB[j][kk]);
                if (relu) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[i] = relu16B<T>(a[i]);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a[i]),
                                                                            __builtin_bit_cast(bf16x8_t, b[j]), acc[i][j], 0, 0, 0);
            }
        }
    };

    issue(0, 0);
    issue(nch > 1 ? 1 : 0, 1);
    // unrolled by the 3 stages so every stage index (LDS immediate offsets, DMA bases) is a compile-time constant
    for (int c0 = 0; c0 < nch; c0 += G_NSTAGE) {
#pragma unroll
        for (int st = 0; st < G_NSTAGE; ++st) {
            const int c = c0 + st;
            if (c < nch) {
                // chunk c has landed once at most the 3 loads of chunk c+1 are outstanding; the barrier then (a)
                // publishes every wave's pieces of chunk c and (b) guarantees all waves finished reading the stage
                // the next prefetch overwrites (chunk c-1)
                asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue(c + 2 < nch ? c + 2 : nch - 1, (st + 2) % G_NSTAGE);
                compute(c, st);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // trailing (redundant) prefetches must land before LDS is reused
    __syncthreads();
    nt_epilogue<T, CfgG>(p, acc, lds, m0, n0, tid, NtScatter{p.gmap, p.gmap_st ? (size_t)p.gmap_st : (size_t)p.N, p.gmap_sc, p.scatter_scale, p.N});
}

// ================================================================================================ TN
// Transposing stage: each thread owns a square block (8x8 bf16 / 4x4 fp32) of the [rows m][cols] source
// tile, loads it with coalesced 16-byte row reads, transposes it in registers and writes 16-byte rows of
// the [col][m] LDS image, so fragment reads are the same ds_read_b128 as in the NT kernel.
#define TN_RB 144
#define TN_TILE (128 * TN_RB)
#define TN_STAGE (2 * TN_TILE)
template <typename T> struct TnStage;
template <> struct TnStage<bf16_t> {
    static constexpr int MC = 64;  // contraction rows per chunk
    uint4 r[8];
    // per-lane source pointer of row 0 of this thread's 8x8 block column (chunk-invariant), set once by init()
    const char* src;
    size_t ld;
    int mrow;   // first row of the block inside a chunk
    // bias gradient (GemmTN::colsum): sums of this thread's 8 columns of D over the rows it has staged, taken from the REGISTERS on
    // their way into LDS (zero-filled rows included as zeros) and reduced over the 8 threads of a column block at the end.  An earlier
    // version re-read the staged tile from LDS under a partial exec mask; at M = 40,000 (six chunks per slice) one block in a few
    // hundred came back with a wrong sum for 16 columns (lanes 48..63 of wave 0) while the MFMAs fed from the same tile were right
    // -- not understood; the registers are what the MFMAs' operand is made of, and this form costs fewer instructions.
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool docs = false;
    __device__ inline void init(const GemmTN& p, int tid, int n0, int k0) {
        const int op = tid >> 7, b = tid & 127, mb = b & 7, nb = b >> 3;
        const char* base = (const char*)(op == 0 ? p.D : p.A);
        ld = (size_t)(op == 0 ? p.ldd : p.lda) * 2;
        const int lim = (op == 0 ? p.N : p.K) - 8;
        const int col = min((op == 0 ? n0 : k0) + nb * 8, lim);
        mrow = mb * 8;
        src = base + (size_t)col * 2;
    }
    // load() only issues the global loads (nothing consumes them) so the wait lands in store().  Branch-free on
    // purpose: loads inside a conditional block make hipcc wait vmcnt(0) at the join (measured 188 -> 225 us).
    __device__ inline void load(const GemmTN& p, int tid, int mc, int n0, int k0) {
        const int m0 = mc + mrow;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *(const uint4*)(src + (size_t)min(m0 + j, p.M - 1) * ld);
    }
    // 8x8 bf16 transpose in registers: output word q of column c packs rows 2q, 2q+1 -> one v_perm_b32 each
    // (v_perm_b32 D = bytes of {S0,S1}; selector values 0-3 pick S1's bytes, 4-7 pick S0's)
    __device__ inline void store(const GemmTN& p, char* stage, int tid, int mc, int n0, int k0) {
        const int op = tid >> 7, b = tid & 127, mb = b & 7, nb = b >> 3;
        char* tile = stage + op * TN_TILE;
        const bool cok = (op == 0 ? n0 : k0) + nb * 8 < (op == 0 ? p.N : p.K);
        if (!cok || mc + MC > p.M) {   // only edge tiles pay for zero-filling (VALU only: safe to branch around)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = (cok && mc + mb * 8 + j < p.M) ? r[j] : zero4();
        }
        const uint32_t* w = (const uint32_t*)r;  // w[j*4 + q] = row j, column pair q
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            uint32_t o[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t a = w[(2 * q) * 4 + (c >> 1)], bb = w[(2 * q + 1) * 4 + (c >> 1)];
                o[q] = __builtin_amdgcn_perm(bb, a, (c & 1) ? 0x07060302u : 0x05040100u);
            }
            if (docs) {   // wave-uniform.  o[q] = rows 2q, 2q+1 of column c: cs[c] += lo * 1 + hi * 1 (fp32 accumulate)
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(cs[c]) : "v"(o[q]), "v"(0x3f803f80u));
            }
            *(uint4*)(tile + (nb * 8 + c) * TN_RB + mb * 16) = make_uint4(o[0], o[1], o[2], o[3]);
        }
    }
    // column sums -> colsum[n0 + nb * 8 + c]: butterfly over the 8 threads (mb = lane & 7) of a column block, one atomic per column
    __device__ inline void flush_colsum(const GemmTN& p, int tid, int n0) {
        const int b = tid & 127, mb = b & 7, nb = b >> 3;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float v = cs[c];
            v += __shfl_xor(v, 1, WAVE);
            v += __shfl_xor(v, 2, WAVE);
            v += __shfl_xor(v, 4, WAVE);
            if (mb == 0 && n0 + nb * 8 + c < p.N) unsafeAtomicAdd(p.colsum + n0 + nb * 8 + c, v);
        }
    }
};
template <> struct TnStage<float> {
    static constexpr int MC = 32;
    float4 r[2][4];
    float cs[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of this thread's 4 columns of D (see TnStage<bf16_t>)
    bool docs = false;
    __device__ inline void init(const GemmTN&, int, int, int) {}
    __device__ inline void load(const GemmTN& p, int tid, int mc, int n0, int k0) {
        const int mb = tid & 7, nb = tid >> 3;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            const char* base = (const char*)(op == 0 ? p.D : p.A);
            const size_t ld = (size_t)(op == 0 ? p.ldd : p.lda) * 4;
            const int lim = (op == 0 ? p.N : p.K) - 4;
            const int col = min((op == 0 ? n0 : k0) + nb * 4, lim);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = min(mc + mb * 4 + j, p.M - 1);
                r[op][j] = *(const float4*)(base + (size_t)m * ld + (size_t)col * 4);
            }
        }
    }
    __device__ inline void store(const GemmTN& p, char* stage, int tid, int mc, int n0, int k0) {
        const int mb = tid & 7, nb = tid >> 3;
#pragma unroll
        for (int op = 0; op < 2; ++op) {
            char* tile = stage + op * TN_TILE;
            const bool cok = (op == 0 ? n0 : k0) + nb * 4 < (op == 0 ? p.N : p.K);
            const bool relu = op == 1 && p.relu_a;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 v = (cok && mc + mb * 4 + j < p.M) ? r[op][j] : make_float4(0.f, 0.f, 0.f, 0.f);
                if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                r[op][j] = v;
                if (op == 0 && docs) { cs[0] += v.x; cs[1] += v.y; cs[2] += v.z; cs[3] += v.w; }
            }
            *(float4*)(tile + (nb * 4 + 0) * TN_RB + mb * 16) = make_float4(r[op][0].x, r[op][1].x, r[op][2].x, r[op][3].x);
            *(float4*)(tile + (nb * 4 + 1) * TN_RB + mb * 16) = make_float4(r[op][0].y, r[op][1].y, r[op][2].y, r[op][3].y);
            *(float4*)(tile + (nb * 4 + 2) * TN_RB + mb * 16) = make_float4(r[op][0].z, r[op][1].z, r[op][2].z, r[op][3].z);
            *(float4*)(tile + (nb * 4 + 3) * TN_RB + mb * 16) = make_float4(r[op][0].w, r[op][1].w, r[op][2].w, r[op][3].w);
        }
    }
    __device__ inline void flush_colsum(const GemmTN& p, int tid, int n0) {
        const int mb = tid & 7, nb = tid >> 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = cs[c];
            v += __shfl_xor(v, 1, WAVE);
            v += __shfl_xor(v, 2, WAVE);
            v += __shfl_xor(v, 4, WAVE);
            if (mb == 0 && n0 + nb * 4 + c < p.N) unsafeAtomicAdd(p.colsum + n0 + nb * 4 + c, v);
        }
    }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTN p, int rows_per_slice) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int MC = TnStage<T>::MC;
    constexpr int BN = 128;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, wm = wv >> 1, wn = wv & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_k = (p.K + BN - 1) / BN;
    const int lin = xcd_remap(blockIdx.x, gridDim.x);  // lin = slice * (tiles_n*tiles_k) + tile
    const int tile = lin % (tiles_n * tiles_k);
    const int n0 = (tile / tiles_k) * BN, k0 = (tile % tiles_k) * BN;
    const int mbeg = (lin / (tiles_n * tiles_k)) * rows_per_slice;
    const int mend = (mbeg + rows_per_slice < p.M) ? mbeg + rows_per_slice : p.M;
    const bool relu_b = sizeof(T) == 2 && p.relu_a;   // bf16: ReLU of the activation operand on its fragments

    // next row block of a tile that touches the scale: whole inactive tiles are skipped at once, eight at a time where the range
    // allows (a launch for a scale no ray reaches used to spend ~40 us walking the mask in MC-row steps)
    auto next_valid = [&](int m) {
        if (p.tile_mask && p.skip_bit >= 0) {
            const uint64_t bit8 = 0x0101010101010101ull << p.skip_bit;
            while (m < mend) {
                const int t = m / SCENERF_TILE_ROWS;
                if ((t & 7) == 0 && (t + 8) * SCENERF_TILE_ROWS <= mend && !(*(const uint64_t*)(p.tile_mask + t) & bit8)) {
                    m = (t + 8) * SCENERF_TILE_ROWS;
                    continue;
                }
                if ((p.tile_mask[t] >> p.skip_bit) & 1u) break;
                m = (t + 1) * SCENERF_TILE_ROWS;
            }
        }
        return m;
    };

    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    TnStage<T> st;
    st.init(p, tid, n0, k0);
    // bias gradient: the k-tile-0 workgroups also sum the columns of D (bf16: the 128 threads that stage D; fp32: every thread stages both)
    st.docs = p.colsum != nullptr && k0 == 0 && (sizeof(T) == 4 || tid < 128);
    int c = next_valid(mbeg);
    if (c >= mend) return;  // uniform
    st.load(p, tid, c, n0, k0);
    st.store(p, lds, tid, c, n0, k0);
    __syncthreads();
    int buf = 0;
    while (c < mend) {
        const int nx = next_valid(c + MC);
        if (nx < mend) st.load(p, tid, nx, n0, k0);
        const char* Dt = lds + buf * TN_STAGE;
        const char* At = Dt + TN_TILE;
#pragma unroll
        for (int kk = 0; kk < MC / 16; ++kk) WaveMma<T, 2, TN_RB>::step(acc, Dt, At, kk, lane, wm, wn, relu_b);
        if (nx < mend) st.store(p, lds + (buf ^ 1) * TN_STAGE, tid, nx, n0, k0);
        __syncthreads();
        buf ^= 1;
        c = nx;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (n < p.N && k < p.K) unsafeAtomicAdd(p.out + (size_t)n * p.ldo + k, acc[i][j][r]);
            }
        }
    if (st.docs) st.flush_colsum(p, tid, n0);
}

// ================================================================================================ launchers
// FLOPs actually issued (profile mode only; synchronises to read the scale-activity mask): row tiles whose
// mask bit is clear skip that segment (or the whole tile when skip_bit is set), so dense 2*M*N*K would overstate.
static std::vector<uint8_t> read_mask(const uint8_t* d, int n, hipStream_t s) {
    std::vector<uint8_t> h(n, 0x1f);
    if (d && hipMemcpyAsync(h.data(), d, n, hipMemcpyDeviceToHost, s) == hipSuccess) (void)hipStreamSynchronize(s);
    return h;
}
static double nt_issued_flops(const GemmNT& p, hipStream_t s) {
    int tiles = cdiv(p.M, SCENERF_TILE_ROWS);
    if (!p.tile_mask) {
        double k = p.K1;
        for (int i = 0; i < p.nseg; ++i) k += p.seg_len[i];
        return 2.0 * p.M * (double)p.N * k;
    }
    std::vector<uint8_t> m = read_mask(p.tile_mask, tiles, s);
    double f = 0;
    for (int t = 0; t < tiles; ++t) {
        int rows = p.M - t * SCENERF_TILE_ROWS < SCENERF_TILE_ROWS ? p.M - t * SCENERF_TILE_ROWS : SCENERF_TILE_ROWS;
        if (p.skip_bit >= 0 && !((m[t] >> p.skip_bit) & 1)) continue;
        double k = p.K1;
        for (int i = 0; i < p.nseg; ++i)
            if ((m[t] >> i) & 1) k += p.seg_len[i];
        f += 2.0 * rows * (double)p.N * k;
    }
    return f;
}
static double tn_issued_flops(const GemmTN& p, hipStream_t s) {
    if (!p.tile_mask || p.skip_bit < 0) return 2.0 * p.M * (double)p.N * p.K;
    int tiles = cdiv(p.M, SCENERF_TILE_ROWS);
    std::vector<uint8_t> m = read_mask(p.tile_mask, tiles, s);
    double f = 0;
    for (int t = 0; t < tiles; ++t) {
        int rows = p.M - t * SCENERF_TILE_ROWS < SCENERF_TILE_ROWS ? p.M - t * SCENERF_TILE_ROWS : SCENERF_TILE_ROWS;
        if ((m[t] >> p.skip_bit) & 1) f += 2.0 * rows * (double)p.N * p.K;
    }
    return f;
}

template <typename T, typename CF> static int launch_nt_t(const GemmNT& p, hipStream_t s) {
    SRF_ONCE_PER_DEVICE(SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, CF>, hipFuncAttributeMaxDynamicSharedMemorySize, CF::LDS)));
    dim3 grid(cdiv(p.M, BM) * (p.ms_n > 0 ? p.ms_t0[GEMM_MAX_SEG] : cdiv(p.N, CF::BN)));
    double flops = 0;
    if (srf_prof_on()) {
        if (p.ms_n > 0) {   // sum over the scales, each with its own activity bit
            for (int i = 0; i < p.ms_n; ++i) {
                if (!p.ms_gmap[i]) continue;
                GemmNT q = p;
                q.ms_n = 0; q.N = p.ms_C[i]; q.skip_bit = i;
                flops += nt_issued_flops(q, s);
            }
        } else flops = nt_issued_flops(p, s);
    }
    SrfLaunchScope ps(s, p.name, flops, 0);
    gemm_nt_kernel<T, CF><<<grid, CF::NT, CF::LDS, s>>>(p);
    SRF_LAUNCH_CHECK(p.name);
    return 0;
}

static int launch_nt_glds(const GemmNT& p, hipStream_t s) {
    SRF_ONCE_PER_DEVICE(SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_glds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CfgG::LDS)));
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, CfgG::BN));
    double flops = 0;
    if (srf_prof_on()) flops = nt_issued_flops(p, s);
    SrfLaunchScope ps(s, p.name, flops, 0);
    gemm_nt_glds_kernel<<<grid, CfgG::NT, CfgG::LDS, s>>>(p);
    SRF_LAUNCH_CHECK(p.name);
    return 0;
}

// every dynamic-LDS attribute of this file, for the current device (scenerf_hip_prepare: before a hipGraph capture)
int gemm_prepare() {
    SRF_ONCE_PER_DEVICE(
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16_t, CfgS>, hipFuncAttributeMaxDynamicSharedMemorySize, CfgS::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<float, CfgS>, hipFuncAttributeMaxDynamicSharedMemorySize, CfgS::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16_t, CfgM>, hipFuncAttributeMaxDynamicSharedMemorySize, CfgM::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<float, CfgM>, hipFuncAttributeMaxDynamicSharedMemorySize, CfgM::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<bf16_t, CfgW>, hipFuncAttributeMaxDynamicSharedMemorySize, CfgW::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_kernel<float, CfgW>, hipFuncAttributeMaxDynamicSharedMemorySize, CfgW::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_nt_glds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CfgG::LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN_STAGE));
        SRF_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN_STAGE)));
    return 0;
}

int launch_gemm_nt(int precision, const GemmNT& p, hipStream_t s) {
    SRF_CHECK((p.W || p.ms_n > 0) && p.M > 0 && p.N > 0, "%s: bad operands", p.name);
    SRF_CHECK(p.K1 % 16 == 0 && p.N % 8 == 0, "%s: K1=%d must be a multiple of 16, N=%d of 8", p.name, p.K1, p.N);
    SRF_CHECK(p.K1 == 0 || p.A1, "%s: A1 NULL", p.name);
    SRF_CHECK(p.nseg == 0 || p.A2, "%s: A2 NULL", p.name);
    for (int i = 0; i < p.nseg; ++i) SRF_CHECK(p.seg_len[i] % 16 == 0 && p.seg_off[i] % 8 == 0, "%s: segment %d misaligned", p.name, i);
    SRF_CHECK(p.scatter_scale >= 0 ? ((p.gmap || p.ms_n > 0) && p.tap_texel && p.tap_weight) : (p.out != nullptr), "%s: missing output", p.name);
    // wide tile when the output is a full 512-column hidden layer and there are enough row tiles to fill the chip
    // hidden layers (N = 512) with enough row tiles to fill the chip: 128 x 256 tiles, two 8-wave workgroups per CU
    // (measured 466 TF/s vs 330 for the 128 x 512 single-workgroup tile and 377 for 128 x 128 at K = 512: with two
    // resident workgroups one's epilogue / LDS refill overlaps the other's MFMAs)
    if (p.ms_n > 0) {   // multi-scale scatter: 128-column tiles of the register-staged kernel only
        SRF_CHECK(p.scatter_scale >= 0 && p.ms_n <= GEMM_MAX_SEG && p.ms_t0[GEMM_MAX_SEG] > 0, "%s: bad multi-scale setup", p.name);
        return precision ? launch_nt_t<bf16_t, CfgS>(p, s) : launch_nt_t<float, CfgS>(p, s);
    }
    const bool big = (p.N % 256 == 0) && cdiv(p.M, BM) >= 192;
    if (precision && (p.force_tile == 4 || (p.force_tile == 0 && big))) return launch_nt_glds(p, s);
    if (p.force_tile == 3 || p.force_tile == 4 || (p.force_tile == 0 && big)) return precision ? launch_nt_t<bf16_t, CfgM>(p, s) : launch_nt_t<float, CfgM>(p, s);
    if (p.force_tile == 2 && p.N == 512) return precision ? launch_nt_t<bf16_t, CfgW>(p, s) : launch_nt_t<float, CfgW>(p, s);
    return precision ? launch_nt_t<bf16_t, CfgS>(p, s) : launch_nt_t<float, CfgS>(p, s);
}

template <typename T> static int launch_tn_t(const GemmTN& p, hipStream_t s) {
    SRF_ONCE_PER_DEVICE(SRF_HIP(hipFuncSetAttribute((const void*)gemm_tn_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TN_STAGE)));
    constexpr int MC = TnStage<T>::MC;
    int tiles = cdiv(p.N, 128) * cdiv(p.K, 128);
    int chunks = cdiv(p.M, MC);
    const int target = 512;   // workgroups the M-split aims for: 256 CUs x 2 resident workgroups = one full wave, no tail (measured best)
    int slices = target / tiles;
    if (slices < 1) slices = 1;
    if (slices > chunks) slices = chunks;
    int cps = cdiv(chunks, slices);  // chunks per slice
    // keep slices aligned to the 128-row mask granularity
    int rows = cps * MC;
    rows = cdiv(rows, SCENERF_TILE_ROWS) * SCENERF_TILE_ROWS;
    slices = cdiv(p.M, rows);
    dim3 grid(tiles * slices);
    double flops = 0;
    if (srf_prof_on()) flops = tn_issued_flops(p, s);
    SrfLaunchScope ps(s, p.name, flops, 0);
    gemm_tn_kernel<T><<<grid, 256, 2 * TN_STAGE, s>>>(p, rows);
    SRF_LAUNCH_CHECK(p.name);
    return 0;
}

int launch_gemm_tn(int precision, const GemmTN& p, hipStream_t s) {
    SRF_CHECK(p.D && p.A && p.out && p.M > 0 && p.N > 0 && p.K > 0, "%s: bad operands", p.name);
    SRF_CHECK(p.N % 8 == 0 && p.K % 8 == 0, "%s: N=%d and K=%d must be multiples of 8", p.name, p.N, p.K);
    if (precision && wgrad_tr_applicable(p)) return launch_wgrad_tr(p, s);
    return precision ? launch_tn_t<bf16_t>(p, s) : launch_tn_t<float>(p, s);
}
