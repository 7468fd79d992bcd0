// Weight-gradient GEMM for gfx950, bf16 operands:  out[n][k] += sum_m D[m][n] * act(A[m][k])   (contraction over the ROWS of two
// row-major activation matrices; resnetfc.py's fc_0 / fc_1 weights: dW = dOut^T relu(In)).
//
// Both MFMA operands are "transposed" accesses (lane = column, 8 consecutive rows per lane).  The first wgrad kernel
// (gemm_tn_kernel, gemm.hip) transposes 8x8 blocks in registers with v_perm on the way into LDS; this one stages nothing in
// registers: row-major [16 rows][256 columns] tiles of D and A go straight from global memory into LDS (global_load_lds, one
// 1-KiB piece = two rows), and the fragments come out with gfx950's transposing LDS read, ds_read_b64_tr_b16: within a group
// of 16 lanes, lane 4r + c supplies the address of 4 consecutive columns of row r, and lane i receives rows 0..3 of column i
// (probed on hardware: tools/ubench/tr_read_probe.hip) -- two of them give a lane its 8 consecutive rows.  Rows are 512 B in
// LDS; the 64-byte chunk index is XORed with (row & 3) on the DMA's source side, which makes the four rows of a transposing
// read hit four different bank quarters.
//
// Same wave specialisation as fused.hip: 8 consumer waves (2 x 4 over a 256 x 256 output tile, wave tile 128 x 64, 128 fp32
// accumulators per lane) and 4 producer waves (one per SIMD) streaming through an 8-stage ring with a counted vmcnt; one raw
// s_barrier per 16 rows.  The M range is split over workgroups; partial tiles are added with fp32 atomics.  ReLU is applied
// to the fragments (v_pk_max_i16); the bias gradient (column sums of D) is accumulated from the D fragments by the workgroups
// of the first k tile.
#include "gemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_w;
typedef __attribute__((ext_vector_type(16))) float f32x16_w;
typedef unsigned u32x2_w __attribute__((ext_vector_type(2)));

#define W_TILE 256                 // output tile: 256 (n) x 256 (k)
#ifndef W_BM
#define W_BM 32                    // rows per step (W_BM / 16 MFMA k-steps)
#endif
#define W_ROW 512                  // LDS row: 256 bf16
#define W_HALF (W_BM * W_ROW)      // one operand's part of a stage
#define W_STAGE (2 * W_HALF)
#define W_NST (8 * 16 / W_BM)      // ring depth: 128 KiB in all
#define W_LDS (W_NST * W_STAGE)    // 131072
#define W_PPP (W_BM / 8)           // pieces per producer per operand per stage (a piece = two rows; four producers)
#define W_THREADS 768
#ifndef W_STAGGER
#define W_STAGGER 1
#endif

#define W_MAXPROB 8
struct WgradProblem {
    const char* D;       // [M][ldd] bf16
    const char* A;       // [M][lda] bf16
    float* out;          // [N][ldo] fp32, atomically accumulated
    float* colsum;       // optional [N]: += sum_m D[m][n]
    int ldd2, lda2;      // row strides in BYTES
    int ldo, relu_a;
};
#define W_MAXWG 512
struct WgradArgs {
    WgradProblem prob[W_MAXPROB];   // same M for all; N and K (multiples of 256) may differ.  The weight gradients of a backward pass
                                    // share one launch: one atomic flush per workgroup for all of them instead of one per GEMM
    unsigned short tiles_n[W_MAXPROB], tiles_k[W_MAXPROB];
    // workgroup id -> problem (bits 12..15) | tile (bits 6..11) | split (bits 0..5); 0xffff = idle.  Host-built so that the tiles
    // of one (problem, split) get ids on the same XCD, back to back
    unsigned short map[W_MAXWG];
    int M, rows_per_split;
    const char* zero;    // >= 1 KiB of zeros (rows past M)
#ifdef W_CYC
    unsigned long long* cyc;
#endif
};

__device__ static inline void w_glds16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}
// same, 64-bit per-lane address (tail rows: some lanes read the zero page)
__device__ static inline void w_glds16v(const void* vaddr, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(vaddr), "s"(lds_wave_base)
                 : "memory");
}
__device__ static inline u32x2_w w_tr_read(unsigned addr) {
    u32x2_w v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

__device__ static inline uint32_t w_max16(uint32_t w, uint32_t f) {
    typedef short w_short2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(w_short2, w), __builtin_bit_cast(w_short2, f)));
}

__global__ __launch_bounds__(W_THREADS) void wgrad_tr_kernel(WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    // XCD-aware placement: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the tiles of one
    // M split -- which read the same rows of D and of A -- are given ids that land on the same XCD, back to back: every row slab
    // then comes from HBM once and from that L2 for the other tiles
    // XCD-aware placement (host-built map): consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so
    // the tiles of one M split -- which read the same rows of D and of A -- get ids that land on the same XCD, back to back: every
    // row slab then comes from HBM once and from that L2 for the other tiles
    const unsigned code = p.map[blockIdx.x];
    if (code == 0xffffu) return;
    const int pi = code >> 12, tile = (code >> 6) & 63, split = code & 63;
    const WgradProblem& P = p.prob[pi];
    const int tn = p.tiles_n[pi];
    const int n0 = (tile % tn) * W_TILE, k0 = (tile / tn) * W_TILE;
    const int kt = tile / tn, tk = p.tiles_k[pi];   // this tile's k index, k tiles of the problem
    const int mb = split * p.rows_per_split;
    const int me = min(p.M, mb + p.rows_per_split);
    const int steps = (me - mb + W_BM - 1) / W_BM;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    if (steps <= 0) return;

    if (wvu >= 8) {
        // ================================================================================ producers
        // a piece = two 512-byte rows of one operand's tile; producer q moves pieces 2q, 2q+1 of D and of A.  Lane L: row 2 piece +
        // L / 32 of the stage, physical 16-byte slot L & 31 of that row, fetching logical chunk (slot / 4) ^ (row & 3)
        const int q = wvu - 8;
        unsigned offD[W_PPP], offA[W_PPP];
#pragma unroll
        for (int u = 0; u < W_PPP; ++u) {
            const int row = 2 * (W_PPP * q + u) + (lane >> 5);
            const int chunk = ((lane & 31) >> 2) ^ (row & 3);
            offD[u] = (unsigned)row * P.ldd2 + n0 * 2 + chunk * 64 + (lane & 3) * 16;
            offA[u] = (unsigned)row * P.lda2 + k0 * 2 + chunk * 64 + (lane & 3) * 16;
        }
        auto issue = [&](int c) {
#ifdef W_VAR_SAMEROWS
            const int m = (c & 7) * W_BM;      // timing experiment: every step re-reads the same 256 rows (L2-resident)
#else
            const int m = mb + c * W_BM;
#endif
            const unsigned sb = lds0 + (c % W_NST) * W_STAGE;
            if (m + W_BM <= p.M) {
                const char* dD = P.D + (size_t)m * P.ldd2;
                const char* dA = P.A + (size_t)m * P.lda2;
#pragma unroll
                for (int u = 0; u < W_PPP; ++u) {
                    w_glds16(dD, offD[u], __builtin_amdgcn_readfirstlane(sb + (W_PPP * q + u) * 1024));
                    w_glds16(dA, offA[u], __builtin_amdgcn_readfirstlane(sb + W_HALF + (W_PPP * q + u) * 1024));
                }
            } else {   // the last rows of the matrix: rows past M come from the zero page
#pragma unroll
                for (int u = 0; u < W_PPP; ++u) {
                    const int row = 2 * (W_PPP * q + u) + (lane >> 5);
                    const bool ok = m + row < p.M;
                    const char* gD = ok ? P.D + (size_t)m * P.ldd2 + offD[u] : p.zero + (lane & 31) * 16;
                    const char* gA = ok ? P.A + (size_t)m * P.lda2 + offA[u] : p.zero + (lane & 31) * 16;
                    w_glds16v(gD, __builtin_amdgcn_readfirstlane(sb + (W_PPP * q + u) * 1024));
                    w_glds16v(gA, __builtin_amdgcn_readfirstlane(sb + W_HALF + (W_PPP * q + u) * 1024));
                }
            }
        };
#pragma unroll 1
        for (int c = 0; c < W_NST - 1 && c < steps; ++c) issue(c);
#pragma unroll 1
        for (int c = 0; c < steps; ++c) {
            // chunk c has landed once at most the loads of chunks c+1 .. c+W_NST-2 (2 W_PPP per producer each) are outstanding
            static_assert((W_NST - 2) * 2 * W_PPP == 24 || (W_NST - 2) * 2 * W_PPP == 16, "update the counted wait");
            if (c + W_NST - 2 < steps) {
                if ((W_NST - 2) * 2 * W_PPP == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            }
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail
            __builtin_amdgcn_s_barrier();
            if (c + W_NST - 1 < steps) issue(c + W_NST - 1);   // into the stage the consumers read in step c - 1
        }
        return;
    }

    // ==================================================================================== consumers
    const int wn = wv >> 2, wk = wv & 3;   // wave tile: n in [128 wn, +128), k in [64 wk, +64)
    // transposing-read addresses: lane l = 16 g + 4 r + c supplies (row 8 (g >> 1) + r [+ 4 for the second half], 4 columns at
    // 16 (g & 1) + 4 c of the 32-column tile) and receives rows 0..3 [4..7] of column l & 15 (+ 16 (g & 1)): lane l & 31 = column
    const int g = lane >> 4, r = (lane >> 2) & 3, cq = lane & 3;
    const int rowoff = (8 * (g >> 1) + r) * W_ROW;
    const int inchunk = (16 * (g & 1) + 4 * cq) * 2;
    unsigned adD[4], adA[2];   // per 32-column tile: byte offset inside a stage of the first read (second: + 4 rows)
#pragma unroll
    for (int i = 0; i < 4; ++i) adD[i] = rowoff + (((wn * 4 + i) ^ r) << 6) + inchunk;
#pragma unroll
    for (int j = 0; j < 2; ++j) adA[j] = W_HALF + rowoff + (((wk * 2 + j) ^ r) << 6) + inchunk;

    f32x16_w acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};   // column sums of D for column n0 + 128 wn + 32 i + (lane & 31), this lane's 8 rows
    // bias gradient (column sums of D): every k tile of a column block sees the same D fragments, so they take turns, step by step.
    // The tiles of one (problem, M split) then do the SAME work per step: with the sums on the first k tile only, that tile fell
    // behind the ones it shares its operand rows with, out of reach of the XCD's L2 (4 MiB hold about nine steps of an XCD's operand
    // stream), and every shared row was fetched twice: 3.39 GB per launch instead of 2.40 GB, 712-740 us instead of 637-650 us
    // (r02_d).  Holding the tiles together by force instead (a progress record per tile, producers waiting for the slowest) also
    // brings the fetch to 2.4 GB but costs more in stalls than it saves: 1.0-1.1 ms.
    const bool do_cs = P.colsum != nullptr && wk == 0;
    const uint32_t floor2 = P.relu_a ? 0u : 0x80008000u;

    // One step = two k-steps of [12 transposing reads -> 8 MFMAs].  All consumer waves leave the barrier together, so without further
    // arrangement the two waves of a SIMD read at the same time (matrix pipe idle) and then multiply at the same time (LDS idle):
    // 2,200 cycles per step against 1,024 of MFMA work, also with every operand L2-resident (r02_d).  The second wave of each SIMD
    // (waves 4..7: wn = 1) therefore runs half a phase late: it carries the fragments of a step's second k-step across the barrier
    // and multiplies them while the first wave reads -- same registers, same barrier, complementary phases.
    uint4 A4[4], B4[2];
    auto rd = [&](unsigned Sk) {
        u32x2_w fa[4][2], fb[2][2];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[i][0] = w_tr_read(Sk + adD[i]);
            fa[i][1] = w_tr_read(Sk + adD[i] + 4 * W_ROW);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            fb[j][0] = w_tr_read(Sk + adA[j]);
            fb[j][1] = w_tr_read(Sk + adA[j] + 4 * W_ROW);
        }
        // the wait carries the fragments as operands: nothing that uses them may be scheduled above it
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[1][0]), "+v"(fa[1][1]), "+v"(fa[2][0]), "+v"(fa[2][1]), "+v"(fa[3][0]),
                       "+v"(fa[3][1]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1])
                     :
                     : "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) A4[i] = make_uint4(fa[i][0][0], fa[i][0][1], fa[i][1][0], fa[i][1][1]);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // ReLU as a 16-bit integer max with 0 -- or with the most negative value, which changes nothing (no branch in the loop)
            B4[j] = make_uint4(w_max16(fb[j][0][0], floor2), w_max16(fb[j][0][1], floor2), w_max16(fb[j][1][0], floor2),
                               w_max16(fb[j][1][1], floor2));
        }
    };
    auto mm = [&](int c) {   // c: the step the fragments belong to
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_w, A4[i]), __builtin_bit_cast(bf16x8_w, B4[j]),
                                                                    acc[i][j], 0, 0, 0);
        if (do_cs && (c % tk) == kt) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // cs += lo + hi of each bf16 pair (v_dot2c with a pair of ones)
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(cs[i]) : "v"(A4[i].x), "v"(0x3f803f80u));
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(cs[i]) : "v"(A4[i].y), "v"(0x3f803f80u));
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(cs[i]) : "v"(A4[i].z), "v"(0x3f803f80u));
                asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(cs[i]) : "v"(A4[i].w), "v"(0x3f803f80u));
            }
        }
    };
    static_assert(W_BM == 32 || !W_STAGGER, "the staggered loop is written for two k-steps per step");
#ifdef W_CYC
    unsigned long long tb = 0, tr0 = 0, tm0 = 0, tr1 = 0, tm1 = 0, t0, t1;
#define W_T(acc_) do { t1 = __builtin_readcyclecounter(); acc_ += t1 - t0; t0 = t1; } while (0)
#else
#define W_T(acc_) do { } while (0)
#endif
    if (W_STAGGER && wn == 1) {
#pragma unroll 1
        for (int c = 0; c < steps; ++c) {
#ifdef W_CYC
            t0 = __builtin_readcyclecounter();
#endif
            __builtin_amdgcn_s_barrier();
            W_T(tb);
            const unsigned S = lds0 + (c % W_NST) * W_STAGE;
            if (c > 0) mm(c - 1);          // second k-step of the previous step (fragments read before the barrier)
            W_T(tm1);
            rd(S);
            W_T(tr0);
            mm(c);
            W_T(tm0);
            rd(S + 16 * W_ROW);            // in registers before the next barrier releases this stage
            W_T(tr1);
        }
        mm(steps - 1);
    } else {
#pragma unroll 1
        for (int c = 0; c < steps; ++c) {
#ifndef W_CYC
            __builtin_amdgcn_s_barrier();
#endif
            const unsigned S = lds0 + (c % W_NST) * W_STAGE;
#ifdef W_CYC
            t0 = __builtin_readcyclecounter();
            __builtin_amdgcn_s_barrier();
            W_T(tb);
            rd(S); W_T(tr0);
            mm(c); W_T(tm0);
            rd(S + 16 * W_ROW); W_T(tr1);
            mm(c); W_T(tm1);
#else
#pragma unroll
            for (int ks = 0; ks < W_BM / 16; ++ks) {
                rd(S + ks * 16 * W_ROW);
                mm(c);
            }
#endif
        }
    }
#ifdef W_CYC
    if (lane == 0 && (wv == 0 || wv == 4) && blockIdx.x < 64) {
        unsigned long long* o = p.cyc + (blockIdx.x * 2 + (wv >> 2)) * 8;
        o[0] = tb; o[1] = tr0; o[2] = tm0; o[3] = tr1; o[4] = tm1; o[5] = steps;
    }
#endif
    // ---- partial tile -> out (fp32 atomics; C tile layout: row (e & 3) + 8 (e >> 2) + 4 (lane >> 5), column lane & 31)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float* o = P.out + (size_t)(n0 + wn * 128 + i * 32 + 4 * (lane >> 5)) * P.ldo + k0 + wk * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) unsafeAtomicAdd(o + (size_t)((e & 3) + 8 * (e >> 2)) * P.ldo, acc[i][j][e]);
        }
    if (do_cs) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = cs[i] + __shfl_xor(cs[i], 32);   // the two lane halves hold rows 0..7 and 8..15 of the same column
            if (lane < 32) unsafeAtomicAdd(P.colsum + n0 + wn * 128 + i * 32 + lane, t);
        }
    }
}


// Measured alternative (r02_e, not kept): the same kernel without producer waves -- eight waves with 256 registers each, two sets of
// fragments per wave so that the reads of the next k-step are in flight while the current one multiplies (also across the barrier),
// every wave moving its own four 1-KiB pieces per step by DMA.  Correct, and 5 % SLOWER than the twelve-wave form (615-622 us
// against 588 us for the batch): the step is not bound by a wave's own read -> multiply chain.  tools/ubench/mfma_lds.hip has the
// single-wave picture (12 LDS reads around 8 MFMAs: 248 ns against 134 ns for the MFMAs alone with one wave per SIMD, 304 against 251
// with two).

// Used by launch_gemm_tn for the shapes it fits (bf16, N and K multiples of 256, no tile skipping, enough rows).
bool wgrad_tr_applicable(const GemmTN& p, int min_rows) {
    return p.allow_tr && p.N % W_TILE == 0 && p.K % W_TILE == 0 && !p.tile_mask && p.M >= min_rows && p.ldd % 8 == 0 && p.lda % 8 == 0;
}

#ifdef W_CYC
static unsigned long long* g_w_cyc = nullptr;
extern "C" int scenerf_hip_test_wgrad_cyc(unsigned long long* out, int n) {
    if (!g_w_cyc || hipDeviceSynchronize() != hipSuccess) return 1;
    return hipMemcpy(out, g_w_cyc, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess;
}
#endif
int wgrad_prepare() {
    SRF_ONCE_PER_DEVICE(SRF_HIP(hipFuncSetAttribute((const void*)wgrad_tr_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, W_LDS)));
    SRF_CHECK(srf_zero_page(), "wgrad: cannot allocate the zero page");
    return 0;
}

// `count` problems over the same M rows (N, K multiples of 256, possibly different) in one launch
int launch_wgrad_tr_batch(const GemmTN* probs, int count, hipStream_t s) {
    if (int e = wgrad_prepare()) return e;
    const char* zero = srf_zero_page();
    SRF_CHECK(zero, "wgrad batch: no zero page on this device");
    SRF_CHECK(count >= 1 && count <= W_MAXPROB, "wgrad batch: 1..%d problems", W_MAXPROB);
    const GemmTN& p0 = probs[0];
    WgradArgs a;
    int tile_units = 0;
    double flops = 0;
    for (int i = 0; i < count; ++i) {
        const GemmTN& p = probs[i];
        SRF_CHECK(p.M == p0.M && wgrad_tr_applicable(p, 1), "wgrad batch: problems must share M and fit the kernel");
        a.prob[i] = {(const char*)p.D, (const char*)p.A, p.out, p.colsum, p.ldd * 2, p.lda * 2, p.ldo, p.relu_a};
        a.tiles_n[i] = (unsigned short)(p.N / W_TILE);
        a.tiles_k[i] = (unsigned short)(p.K / W_TILE);
        SRF_CHECK((p.N / W_TILE) * (p.K / W_TILE) <= 64, "wgrad batch: at most 64 tiles per problem");
        tile_units += (p.N / W_TILE) * (p.K / W_TILE);
        flops += 2.0 * p.M * (double)p.N * p.K;
    }
    a.M = p0.M;
    a.zero = zero;
    const int wg_target = 256;   // one workgroup per CU (128 KiB of LDS)
    int splits = wg_target / tile_units > 1 ? wg_target / tile_units : 1;
    if (splits > 64) splits = 64;
    if (splits > p0.M / 1024) splits = p0.M / 1024 > 1 ? p0.M / 1024 : 1;   // few rows: fewer, longer splits (each split flushes 256 KiB per tile)
    int rows = cdiv(cdiv(p0.M, splits), W_BM) * W_BM;   // (a multiple of the step)
    splits = cdiv(p0.M, rows);
    a.rows_per_split = rows;
    // groups (problem, split) dealt round-robin to the 8 XCDs; each XCD's workgroups = its groups' tiles, in order
    int len[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < W_MAXWG; ++i) a.map[i] = 0xffff;
    int g = 0, grid = 0;
    for (int sp = 0; sp < splits; ++sp)
        for (int i = 0; i < count; ++i, ++g) {
            // eight splits: a split per XCD, so that problems sharing an operand (dH_b is D of fc_1.(b-1) and of lin_z.b) can share it in L2
            const int x = splits == 8 ? sp : (g & 7), nt = (probs[i].N / W_TILE) * (probs[i].K / W_TILE);
            for (int t = 0; t < nt; ++t) {
                const int b = (len[x]++) * 8 + x;
                SRF_CHECK(b < W_MAXWG, "wgrad batch: workgroup map overflow");
                a.map[b] = (unsigned short)((i << 12) | (t << 6) | sp);
                if (b + 1 > grid) grid = b + 1;
            }
        }
#ifdef W_CYC
    static unsigned long long* d_cyc = nullptr;
    if (!d_cyc) SRF_HIP(hipMalloc((void**)&d_cyc, 64 * 2 * 8 * 8));
    a.cyc = d_cyc;
    g_w_cyc = d_cyc;
#endif
    SrfLaunchScope ps(s, p0.name, flops, 0);
    wgrad_tr_kernel<<<grid, W_THREADS, W_LDS, s>>>(a);
    SRF_LAUNCH_CHECK("wgrad_tr_kernel");
    return 0;
}

int launch_wgrad_tr(const GemmTN& p, hipStream_t s) { return launch_wgrad_tr_batch(&p, 1, s); }
