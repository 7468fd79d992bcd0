// Feature-map gradient of one ResnetFC pass for gfx950 (bf16): dZ = dH[:, 0:1536] @ Wz (the three lin_z layers read the same gathered
// features z, so dz = sum_b dH_b Wz_b^T = one K = 1536 product), scattered straight into the (H,W,C) map gradients through the
// bilinear taps the forward gather recorded = grid_sampler_2d_backward of reference scenerf/models/utils.py:232-247 (sample_feats_2d),
// per pyramid level of scenerf.py:522-527.
//
// Why its own kernel (it was gemm_nt_kernel with a scatter epilogue: 397 us at the bench shape, 162 us of it the GEMM alone):
//   * the product is HBM-bound by construction -- 472 MB of dH for 61 GFLOP at KITTI's scale mix -- so the goal is to read dH exactly
//     once at memory speed and to keep everything else out of its way.  One workgroup = one 128-row tile (one ray at N = 128) and ALL
//     the scales that tile touches: the accumulators of up to 8 column tiles (256 channels) live in registers, dH streams through once.
//     Tiles that touch only the finest level (80 channels: 3 column tiles) run in a 3-tile instantiation with a third of the registers
//     and four workgroups per CU; the host launches both, each workgroup leaves at once if the tile is the other kernel's;
//   * the K loop stages dH and the weight rows through LDS by DMA (global_load_lds), 64 bytes of K per row and step, a ring of NS
//     stages with ONE barrier per step: NS - 1 steps are in flight while a step is multiplied, NS - 2 while its pieces are waited for (the first version had two
//     stages of 128 bytes and two barriers per step: one step of lookahead = 28 KB in flight per workgroup, and a tile took 70 us for
//     393 KB -- the kernel ran at 2.3 TB/s with the CUs waiting for memory latency; r02_f).  A 1-KiB piece is 16 whole row segments.  (A first version loaded the MFMA fragments straight into registers -- 32 rows x 32 B
//     per wave instruction, i.e. 32 different pages per load, dH rows being 4 KiB apart -- and ran 2.3x SLOWER than the old kernel:
//     address translation, not bandwidth.)  The 16-byte slots of a row are XOR-swizzled on the source side of the DMA;
//   * the scatter: a ray's samples walk along an epipolar curve, so the 512 (sample, tap) pairs of a tile hit only ~90 distinct texels.
//     Per tap the texel is constant over runs of consecutive samples, so a wave sums its rows per (tap, run) in registers and sends one
//     atomic per (run, channel) -- see the epilogue.  (Measured dead ends, same results: a hash table texel -> slot with a counting
//     sort and a segmented sum -- exact de-duplication, but three dependent LDS reads per entry: 31k cycles per round; LDS float
//     atomics into a per-texel table -- rows that share a texel serialise on one address: 278k cycles per round.)
//   * measured in round 4, not kept (tools/dfeat_probe.py, finest level alone: 153-159 us per launch, 472 MB = 3.0 TB/s; per tile and
//     workgroup 91-98k cycles of K loop + 25k of epilogue, two workgroups per CU):
//       - producer waves (4 + 4 waves, wgrad.hip's protocol, the multiplying waves never issue a vector-memory instruction; two 8-wave
//         workgroups per CU at 128 registers; eight waves in the scatter): same results, 171 us -- the "46k cycles stalled issuing DMA"
//         of r03 are back-pressure of a saturated memory path (a full round of K loops already moves 4.5-5 TB/s), not an issue limit
//         that other waves could lift;
//       - the second resident workgroup of each CU started half a tile late, so that one streams while the other scatters:
//         152-154 us at 4-5 x 8k cycles of delay, 162 at 8 -- one workgroup alone does not pull the CU's share of the stream;
//       - the scatter without its atomics: 136 us (the epilogue 22k instead of 27k cycles): the walk over (row, tap) pairs, not the
//         atomics, is what the epilogue costs.
//     Kept: the first level's taps requested before the K loop (epilogue 31k -> 25k cycles; step time within noise).
#include "gemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_d;
typedef __attribute__((ext_vector_type(16))) float f32x16_d;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_d;

#define DF_BM 128
#ifdef DF_VAR_NOATOM      // (development: the scatter without its atomics -- results are garbage)
#define DF_ATOM(p_, v_) asm volatile("" ::"v"(p_), "v"(v_))
#else
#define DF_ATOM(p_, v_) unsafeAtomicAdd(p_, v_)
#endif
#define DF_THREADS 256
#define DF_K (3 * SCENERF_D_HIDDEN)        // 1536
#define DF_NCH (DF_K / 16)                 // 96 chunks
#define DF_PA 4                            // dH fragments in flight per wave
#ifndef DF_CG
#define DF_CG 96                           // channels per epilogue round: up to three column tiles (a lane takes channel l and l + 64)
#endif
#define DF_RT (DF_CG / 32)                 // column tiles per epilogue round
#ifndef DF_WGS
#define DF_WGS 2                           // workgroups per CU the register budget is set for
#endif
#define DF_CLD (DF_CG + 4)                 // staged row stride in floats
// LDS (epilogue only): staged tile, taps of the level
#define DF_L_CS 0
#define DF_L_TX (DF_L_CS + DF_BM * DF_CLD * 4)        // 51200: [128 rows][4 taps] texel
#define DF_L_TW (DF_L_TX + 512 * 4)                   //        [128 rows][4 taps] weight
#define DF_L_MISC (DF_L_TW + 512 * 4)
#ifndef DF_KSB
#define DF_KSB 64                                     // bytes of K per row per K-loop step
#endif
#ifndef DF_NS3
#define DF_NS3 5                                      // K-loop ring stages of the 3-tile instantiation (71.7 KB: two workgroups per CU)
#endif
#ifndef DF_NS5
#define DF_NS5 4                                      // ... of the DF_NTB-tile one (73.7 KB at 5 tiles)
#endif
constexpr int df_ns(int ntp) { return ntp <= 3 ? DF_NS3 : DF_NS5; }
#ifndef DF_NTB
#define DF_NTB 5                                       // column tiles per pass of the second instantiation (KITTI's 1/2 level: 160 channels)
#endif
// LDS of an instantiation: its K-loop ring of (128 + 32 NTP)-row stages, or the epilogue tables
#define DF_DUMP 4096                               // behind the ring: 1 KiB per wave where the pieces a wave issues only to keep its count uniform land
constexpr int df_lds(int ntp) { return df_ns(ntp) * (DF_BM + ntp * 32) * DF_KSB + DF_DUMP > DF_L_MISC + 32 ? df_ns(ntp) * (DF_BM + ntp * 32) * DF_KSB + DF_DUMP : DF_L_MISC + 32; }

// 16 bytes per lane, global -> LDS: per-lane 64-bit source address, destination = M0 (wave-uniform LDS address) + 16 * lane
__device__ static inline void df_glds16(const void* src, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(src), "s"(lds_wave_base)
                 : "memory");
}

struct DfeatArgs {
    const void* dH;          // [M][ldh] bf16, the first DF_K columns are read
    int ldh, M;
    unsigned live;           // levels of this launch that have a gradient map
    const uint8_t* tile_mask;
    const int32_t* tap_texel;   // [M][5][4]
    const float* tap_weight;    // [M][5][4]
    const void* W[SCENERF_N_SCALES];   // per level: [C_s][DF_K] bf16 (lin_z weights of the three blocks side by side, transposed)
    float* gmap[SCENERF_N_SCALES];     // NULL: that level needs no gradient
    long st[SCENERF_N_SCALES], sc[SCENERF_N_SCALES];   // element strides of gmap per texel / per channel
    int C[SCENERF_N_SCALES];
    unsigned levels;          // pyramid levels this launch handles (bit s)
    int direct;               // 1: workgroup b IS tile b (a launch whose every active tile has exactly one pass: the finest level alone)
    // direct launches: the tiles from `split_from` on are split over `split_n` workgroups each along K (a workgroup's sum goes out
    // through the same atomics) -- the last, partly filled round of workgroups then takes 1 / split_n of a K loop instead of a whole one
    int split_from, split_n;
};

#ifdef H_CYC   // development build: per-workgroup time stamps of wave 0 (tools/dfeat_probe.py)
__device__ unsigned long long* g_df_cyc = nullptr;
extern "C" int scenerf_hip_test_dfeat_cyc(unsigned long long* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_df_cyc), &ptr, sizeof(ptr)); }
#define DF_STAMP()                                                                                  \
    if (wv == 0 && g_df_cyc && ci < 16) {                                                           \
        if (lane == 0) g_df_cyc[(size_t)blockIdx.x * 16 + ci] = __builtin_amdgcn_s_memtime();       \
        ++ci;                                                                                       \
    }
#else
#define DF_STAMP()
#endif

// NTP = column tiles (32 channels each) whose accumulators a pass keeps in registers
template <int NTP>
__global__ __launch_bounds__(DF_THREADS, DF_WGS) void dfeat_kernel(DfeatArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Work item = (tile, pass): workgroup b takes the b-th pass (NTP column tiles) of the tiles that have work in this launch (a level
    // of this launch's set with a gradient map), found by a prefix scan over the mask bytes.  With one workgroup per tile, the four
    // tiles of a KITTI frame that also touch the 1/4 level (15 column tiles = three passes, ~150 us) set the duration of the whole
    // coarser-level launch (162 us for the 322 tiles that have a coarser level; r02_d); and with blockIdx = tile a launch whose
    // active tiles are every fourth one (tools/dfeat_probe.py's periodic masks) ran on two of the eight XCDs.
    // (a launch has one workgroup per tile; the passes beyond that -- tiles with more than NTP column tiles -- go to the first
    // workgroups as second items)
#ifdef H_CYC
    int ci = 0;
#endif
    // The scan: thread i owns tiles i, i + 256, ...; per group of 256 tiles a wave-level prefix of the pass counts (shuffles), the
    // per-(group, wave) totals through LDS, one barrier; every workgroup also learns the number of items, so only the few that have a
    // second item look again.
    constexpr int DF_SCAN_G = 8;       // groups of DF_THREADS tiles per sweep
    int n_items = 0x7fffffff;
    DF_STAMP()   // entry (before the item scan)
    for (int want = blockIdx.x; want < n_items; want += gridDim.x) {
    int tile = -1, pass0 = 0;
    if (p.direct) {
        // one workgroup per tile and at most one pass per tile: no item search (the scan below costs 12.8k of a tile's 120k cycles; r03)
        n_items = 0;                   // (no second item)
        tile = (int)blockIdx.x < p.split_from ? (int)blockIdx.x : p.split_from + ((int)blockIdx.x - p.split_from) / p.split_n;
        if (tile >= (p.M + DF_BM - 1) / DF_BM || !((unsigned)p.tile_mask[tile] & p.live)) return;
        __syncthreads();
    } else {
        int* const s_wc = (int*)lds;   // [4 g + w] pass count of wave w's tiles in group g; [4 DF_SCAN_G ..] the item found: tile, pass
        const int ntile = (p.M + DF_BM - 1) / DF_BM;
        int base = 0;
        for (int sweep0 = 0; sweep0 < ntile; sweep0 += DF_SCAN_G * DF_THREADS) {
            __syncthreads();           // (previous sweep / previous item: its reads of this memory are done)
            if (tid == 0) s_wc[4 * DF_SCAN_G] = -1;
            int cnt[DF_SCAN_G], incl[DF_SCAN_G];
#pragma unroll
            for (int g = 0; g < DF_SCAN_G; ++g) {
                const int t = sweep0 + g * DF_THREADS + tid;
                cnt[g] = 0;
                if (t < ntile) {
                    const unsigned m = (unsigned)p.tile_mask[t] & p.live;
                    int n = 0;
#pragma unroll
                    for (int sc = 0; sc < SCENERF_N_SCALES; ++sc) n += ((m >> sc) & 1u) ? (p.C[sc] + 31) >> 5 : 0;
                    cnt[g] = (n + NTP - 1) / NTP;
                }
            }
#pragma unroll
            for (int g = 0; g < DF_SCAN_G; ++g) {
                incl[g] = cnt[g];
                if (sweep0 + g * DF_THREADS < ntile) {   // (uniform: groups past the last tile cost nothing)
#pragma unroll
                    for (int d = 1; d < 64; d <<= 1) {
                        const int o = __shfl_up(incl[g], d);
                        if (lane >= d) incl[g] += o;
                    }
                }
                if (lane == 63) s_wc[4 * g + wv] = incl[g];
            }
            __syncthreads();
            int run = base;            // items before (group g, wave 0)
#pragma unroll
            for (int g = 0; g < DF_SCAN_G; ++g) {
                if (sweep0 + g * DF_THREADS >= ntile) break;
                int before = run;
                for (int i = 0; i < wv; ++i) before += s_wc[4 * g + i];
                const int excl = before + incl[g] - cnt[g];
                if (cnt[g] > 0 && want >= excl && want < excl + cnt[g]) {
                    s_wc[4 * DF_SCAN_G] = sweep0 + g * DF_THREADS + tid;
                    s_wc[4 * DF_SCAN_G + 1] = want - excl;
                }
                run += s_wc[4 * g] + s_wc[4 * g + 1] + s_wc[4 * g + 2] + s_wc[4 * g + 3];
            }
            base = run;
            __syncthreads();
            if (s_wc[4 * DF_SCAN_G] >= 0) { tile = s_wc[4 * DF_SCAN_G]; pass0 = s_wc[4 * DF_SCAN_G + 1]; }
            // (keep sweeping: the item count decides whether this workgroup comes back)
        }
        n_items = base;
        __syncthreads();              // the scratch words are part of the first K-loop stage
        if (tile < 0) return;         // no such item: done
    }
    pass0 = __builtin_amdgcn_readfirstlane(pass0);
    tile = __builtin_amdgcn_readfirstlane(tile);
    const int m0 = tile * DF_BM;
    const unsigned mask = __builtin_amdgcn_readfirstlane((unsigned)p.tile_mask[tile] & 31u) & p.levels;
    // this tile's column tiles, level by level (wave-uniform, a few scalar registers): (level, first channel)
    int nt = 0;
#pragma unroll
    for (int s = 0; s < SCENERF_N_SCALES; ++s)
        if (((mask >> s) & 1u) && p.gmap[s]) nt += (p.C[s] + 31) >> 5;
    if (nt == 0) continue;
    auto tile_level = [&](int t, int& s_out, int& c0_out) __attribute__((always_inline)) {
        int acc = 0;
        s_out = 0; c0_out = 0;
#pragma unroll
        for (int s = 0; s < SCENERF_N_SCALES; ++s) {
            const int n = (((mask >> s) & 1u) && p.gmap[s]) ? (p.C[s] + 31) >> 5 : 0;
            if (t >= acc && t < acc + n) { s_out = s; c0_out = (t - acc) * 32; }
            acc += n;
        }
    };

    float* const Cs = (float*)(lds + DF_L_CS);
    int* const s_tx = (int*)(lds + DF_L_TX);
    float* const s_tw = (float*)(lds + DF_L_TW);
    DF_STAMP()   // 0: start

    {   // this workgroup's pass: column tiles [t0, t0 + np)
        const int t0 = pass0 * NTP;
        const int np = min(NTP, nt - t0);
        // ---- K loop: acc[t] (C^T tiles: rows = channels, columns = this wave's 32 rows) += W_t[:, step] x dH[rows, step]
        // stage = [128 rows of dH][KSB] ++ [NTP x 32 weight rows][KSB]; 16-byte slot q of row r sits at q ^ swz(r)
        constexpr int KSB = DF_KSB;                       // bytes of K per row per step
        constexpr int SL = KSB / 16;                      // slots per row
        constexpr int RPP = 1024 / KSB;                   // rows per 1-KiB DMA piece
        constexpr int NPA_ = DF_BM / RPP / 4;             // dH pieces per wave per step
        constexpr int NWP = NTP * 32 / RPP;               // weight pieces per step
        constexpr int NPW_ = (NWP + 3) / 4;               // ... per wave (the last wave may have fewer)
        constexpr int STG = (DF_BM + NTP * 32) * KSB;
        constexpr int NSTEP = DF_K * 2 / KSB;
        constexpr int NS = df_ns(NTP), LA = NS - 1;       // ring stages, steps issued ahead of the one being multiplied
        static_assert(NS * STG + DF_DUMP <= df_lds(NTP), "stages must fit the workgroup's LDS");
        static_assert(NSTEP / 4 > LA && NSTEP % 12 == 0, "the ring is shorter than a quarter of the K loop / K splits in 2, 3, 4");
        // this workgroup's part of the K loop (the whole of it unless the tile is split: DfeatArgs.split_from)
        int s_beg = 0, s_end = NSTEP;
        if (p.direct && (int)blockIdx.x >= p.split_from) {
            const int part = ((int)blockIdx.x - p.split_from) % p.split_n;
            s_beg = part * (NSTEP / p.split_n);
            s_end = s_beg + NSTEP / p.split_n;
        }
        auto swz = [](int r) { return KSB == 256 ? r & 15 : KSB == 128 ? (r >> 1) & 7 : (r >> 2) & 3; };
        static_assert(KSB == 256 || KSB == 128 || KSB == 64, "swizzle is written for 16, 8 or 4 slots per row");
        f32x16_d acc[NTP];
#pragma unroll
        for (int t = 0; t < NTP; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // the taps of the pass's first level, requested NOW: two (texel, weight) pairs per thread stay in registers through the K loop and go
        // to LDS behind it (r04: requested after the loop they cost the epilogue 6-7k of its 32k cycles, every wave waiting for the round trip)
        int s_first, c_first;
        tile_level(t0, s_first, c_first);
        int pf_tx[2];
        float pf_tw[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * DF_THREADS, m = m0 + (i >> 2);
            const size_t o = ((size_t)min(m, p.M - 1) * SCENERF_N_SCALES + s_first) * 4 + (i & 3);
            pf_tx[u] = m < p.M ? p.tap_texel[o] : -1;
            pf_tw[u] = m < p.M ? p.tap_weight[o] : 0.f;
        }
        // DMA sources of this wave's pieces: lane -> (row of the piece, physical slot), fetching logical slot physical ^ swz(row)
        const char* srcA[NPA_];
        const char* srcW[NPW_];
        bool okW[NPW_];    // (wave-uniform) a real piece; the others are issued too, into the dump area: every wave then has the SAME
                           // number of pieces in flight per step and "step s has landed" is ONE s_waitcnt immediate -- the per-count
                           // chain of compares and branches this replaces was a third of the loop's 11.5 scalar instructions per MFMA
                           // (SQ counters, profiles/r03_f_pmc_sq_dfeat.txt)
        {
            const int rp = lane / SL, ps = lane % SL;
#pragma unroll
            for (int i = 0; i < NPA_; ++i) {
                const int r = (wv * NPA_ + i) * RPP + rp;
                srcA[i] = (const char*)p.dH + (size_t)min(m0 + r, p.M - 1) * (p.ldh * 2) + ((ps ^ swz(r)) << 4);
            }
#pragma unroll
            for (int i = 0; i < NPW_; ++i) {
                const int r = (wv * NPW_ + i) * RPP + rp;     // row of the weight stage: tile r / 32, channel r % 32 of it
                const int t = r >> 5;
                int sl, c0;
                tile_level(t0 + min(t, np - 1), sl, c0);
                okW[i] = wv * NPW_ + i < NWP && (((wv * NPW_ + i) * RPP) >> 5) < np;   // (wave-uniform; a piece never straddles tiles)
                srcW[i] = (const char*)p.W[sl] + (size_t)min(c0 + (r & 31), p.C[sl] - 1) * (DF_K * 2) + ((ps ^ swz(r)) << 4);
            }
        }
        const unsigned lds0 = (unsigned)(uintptr_t)lds;
        const unsigned dump = lds0 + NS * STG + wv * 1024;
        auto issue = [&](const int step, const int stage) __attribute__((always_inline)) {
            const unsigned sb = lds0 + stage * STG;
#ifdef DF_VAR_SAME   // (development: every step fetches the first step's bytes again -- cache-resident sources)
            const unsigned ko = 0u * (unsigned)step;
#else
            const unsigned ko = (unsigned)step * KSB;
#endif
            // (DF_VAR_*: development knobs -- which of the two DMA streams the K loop waits for; results are garbage with them)
#pragma unroll
            for (int i = 0; i < NPW_; ++i) {
#ifdef DF_VAR_NOW
                if (step >= s_beg + LA) continue;
#endif
                df_glds16(srcW[i] + ko, __builtin_amdgcn_readfirstlane(okW[i] ? sb + DF_BM * KSB + (wv * NPW_ + i) * 1024 : dump));
            }
#pragma unroll
            for (int i = 0; i < NPA_; ++i) {
#ifdef DF_VAR_NOA
                if (step >= s_beg + LA) continue;
#endif
                df_glds16(srcA[i] + ko, __builtin_amdgcn_readfirstlane(sb + (wv * NPA_ + i) * 1024));
            }
        };
        // "step s has landed": when a wave waits for it, steps up to s + LA - 1 have been issued (step s + LA follows the barrier), so at
        // most the pieces of the LA - 1 younger steps may be outstanding (issued in order, retired in order): NPA_ + NPW_ per step
#define DF_WAIT_STEP() asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LA - 1) * (NPA_ + NPW_) < 63 ? (LA - 1) * (NPA_ + NPW_) : 63) : "memory");
        __syncthreads();   // (the previous pass's epilogue is done with the LDS the stages live in)
#pragma unroll
        for (int q = 0; q < LA; ++q) issue(s_beg + q, q);
        const int frow = 32 * wv + (lane & 31), fh = lane >> 5;
        int stage = 0, stage_in = LA;      // stage of the step being multiplied / of the step being issued
#ifdef H_CYC   // where a step's cycles go (wave 0): waiting for its pieces / at the barrier / issuing the next pieces / reads + MFMAs
        unsigned long long ph_[4] = {0, 0, 0, 0}, pt_ = __builtin_amdgcn_s_memtime();
#define DF_PH(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph_[i] += n_ - pt_; pt_ = n_; }
#else
#define DF_PH(i)
#endif
#pragma unroll 1
        for (int step = s_beg; step < s_end; ++step) {
            if (step + LA < s_end) {
                DF_WAIT_STEP()
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the last LA steps: nothing younger is issued any more)
            }
            // one barrier per step: every wave's pieces of this step have landed, and everyone is done multiplying the previous step --
            // whose stage the pieces issued below overwrite
            DF_PH(0)
            __syncthreads();
            DF_PH(1)
            if (step + LA < s_end) issue(step + LA, stage_in);
            DF_PH(2)
            const char* const sa = lds + stage * STG;
            const char* const sw = sa + DF_BM * KSB;
            if (np == NTP) {
                // the common case, a full pass: every fragment of the step is requested before the first MFMA (straight-line code: the
                // guarded form below compiles to read -> wait -> MFMA six times over, ~120 cycles of LDS latency exposed per MFMA --
                // that, not memory, was 80 % of this kernel: without any DMA at all it still took 187 of 230 us; r03)
                bf16x8_d av[KSB / 32], bv[KSB / 32][NTP];
#pragma unroll
                for (int j = 0; j < KSB / 32; ++j) {
                    av[j] = *(const bf16x8_d*)(sa + frow * KSB + (((2 * j + fh) ^ swz(frow)) << 4));
#pragma unroll
                    for (int t = 0; t < NTP; ++t) {
                        const int wr = t * 32 + (lane & 31);
                        bv[j][t] = *(const bf16x8_d*)(sw + wr * KSB + (((2 * j + fh) ^ swz(wr)) << 4));
                    }
                }
#pragma unroll
                for (int j = 0; j < KSB / 32; ++j)
#pragma unroll
                    for (int t = 0; t < NTP; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv[j][t], av[j], acc[t], 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < KSB / 32; ++j) {
                    const bf16x8_d av = *(const bf16x8_d*)(sa + frow * KSB + (((2 * j + fh) ^ swz(frow)) << 4));
#pragma unroll
                    for (int t = 0; t < NTP; ++t) {
                        if (t < np) {
                            const int wr = t * 32 + (lane & 31);
                            const bf16x8_d bv = *(const bf16x8_d*)(sw + wr * KSB + (((2 * j + fh) ^ swz(wr)) << 4));
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bv, av, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
            stage = stage + 1 == NS ? 0 : stage + 1;
            stage_in = stage_in + 1 == NS ? 0 : stage_in + 1;
            DF_PH(3)
        }
#undef DF_WAIT_STEP
#ifdef H_CYC
        if (wv == 0 && g_df_cyc && lane == 0)
            for (int i = 0; i < 4; ++i) g_df_cyc[(size_t)blockIdx.x * 16 + 12 + i] = ph_[i];
#endif
#undef DF_PH

        DF_STAMP()   // K loop done
        // ---- epilogue: level by level, up to three column tiles (96 channels) per round
        int tb = 0;
        int taps_level = -1;
        while (tb < np) {
            int s, c0;
            tile_level(t0 + tb, s, c0);
            int te = tb;
            {
                int s2, c2;
                while (te < np) {
                    tile_level(t0 + te, s2, c2);
                    if (s2 != s) break;
                    ++te;
                }
            }
            float* const g = p.gmap[s];
            const long gst = p.st[s], gsc = p.sc[s];
            for (int tr = tb; tr < te; tr += DF_RT) {
                __syncthreads();   // (the K loop / the previous round is done with this LDS)
                if (taps_level < 0 && s == s_first) {
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        s_tx[tid + u * DF_THREADS] = pf_tx[u];
                        s_tw[tid + u * DF_THREADS] = pf_tw[u];
                    }
                    taps_level = s;
                } else if (taps_level != s) {
                    for (int i = tid; i < 512; i += DF_THREADS) {
                        const int m = m0 + (i >> 2);
                        const size_t o = ((size_t)min(m, p.M - 1) * SCENERF_N_SCALES + s) * 4 + (i & 3);
                        s_tx[i] = m < p.M ? p.tap_texel[o] : -1;
                        s_tw[i] = m < p.M ? p.tap_weight[o] : 0.f;
                    }
                    taps_level = s;
                }
                DF_STAMP()   // taps requested
                // stage: lane = row (lane & 31) of the wave's 32, quads of 4 consecutive channels 8 q + 4 hi
                const int row = 32 * wv + (lane & 31), hi = lane >> 5;
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    if (t >= tr && t < tr + DF_RT && t < te) {
                        const int cb = (t - tr) * 32;
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(float4*)(Cs + row * DF_CLD + cb + 8 * q + 4 * hi) =
                                make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
                    }
                }
                __syncthreads();
                DF_STAMP()   // tile staged
                // scatter.  A ray's samples are sorted along the ray and their texels advance slowly, so for each of the four taps the
                // texel is constant over runs of consecutive rows: a wave walks its 32 rows in order (eight rows of independent LDS
                // reads in flight: staged values, tap texels, tap weights -- no indirection), keeps one running sum per tap (lanes =
                // channels l and l + 64) and sends ONE atomic per (run, channel) when a tap's texel changes: ~30 runs per tap and ray
                // instead of 128 rows -- 256-byte contiguous runs of the (H,W,C) accumulator.
                int s3, cbase;
                tile_level(t0 + tr, s3, cbase);
                const int nch = min(32 * (min(tr + DF_RT, te) - tr), p.C[s] - cbase);   // valid channels of the round
                const bool ok0 = lane < nch, ok1 = lane + 64 < nch;
                float* const gl = g + (size_t)(cbase + lane) * gsc;
                const long g64 = 64 * gsc;
                int cur[4] = {-1, -1, -1, -1};
                float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f};
                const int r0 = 32 * wv;
#pragma unroll 1
                for (int rb = 0; rb < 32; rb += 8) {
                    float c0v[8], c1v[8];
                    int4 t4[8];
                    float4 w4[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int r = r0 + rb + j;
                        c0v[j] = Cs[r * DF_CLD + lane];
                        c1v[j] = Cs[r * DF_CLD + 64 + (lane & 31)];
                        t4[j] = *(const int4*)(s_tx + r * 4);
                        w4[j] = *(const float4*)(s_tw + r * 4);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int tk[4] = {t4[j].x, t4[j].y, t4[j].z, t4[j].w};
                        const float wk[4] = {w4[j].x, w4[j].y, w4[j].z, w4[j].w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int t_ = __builtin_amdgcn_readfirstlane(tk[k]);
                            if (t_ != cur[k]) {   // wave-uniform
                                if (cur[k] >= 0) {
                                    float* const gp = gl + (size_t)cur[k] * gst;
                                    if (ok0) DF_ATOM(gp, v0[k]);
                                    if (ok1) DF_ATOM(gp + g64, v1[k]);
                                }
                                cur[k] = t_; v0[k] = 0.f; v1[k] = 0.f;
                            }
                            v0[k] = fmaf(wk[k], c0v[j], v0[k]);
                            v1[k] = fmaf(wk[k], c1v[j], v1[k]);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (cur[k] >= 0) {
                        float* const gp = gl + (size_t)cur[k] * gst;
                        if (ok0) DF_ATOM(gp, v0[k]);
                        if (ok1) DF_ATOM(gp + g64, v1[k]);
                    }
                }
                DF_STAMP()   // round scattered
            }
            tb = te;
        }
    }
    }   // items
}

// NTP = column tiles (32 channels each) whose accumulators a pass keeps in registers
int launch_dfeat_scatter(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const uint8_t* tile_mask, const int32_t* tap_texel,
                         const float* tap_weight, int M, const void* dH, float* const gmaps[SCENERF_N_SCALES], hipStream_t s) {
    SRF_ONCE_PER_DEVICE(
        SRF_HIP(hipFuncSetAttribute((const void*)dfeat_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, df_lds(3)));
        SRF_HIP(hipFuncSetAttribute((const void*)dfeat_kernel<DF_NTB>, hipFuncAttributeMaxDynamicSharedMemorySize, df_lds(DF_NTB))));
    DfeatArgs p = {};
    p.dH = dH; p.ldh = 4 * SCENERF_D_HIDDEN; p.M = M;
    p.tile_mask = tile_mask; p.tap_texel = tap_texel; p.tap_weight = tap_weight;
    bool any = false;
    for (int sc = 0; sc < SCENERF_N_SCALES; ++sc) {
        p.W[sc] = w->w_z_t[sc];
        p.gmap[sc] = gmaps[sc];
        p.C[sc] = cfg->map_C[sc];
        p.st[sc] = cfg->map_C[sc]; p.sc[sc] = 1;                                                          // (H,W,C)
        if (cfg->map_chw[sc] == 1) { p.st[sc] = 1; p.sc[sc] = (long)cfg->map_H[sc] * cfg->map_W[sc]; }       // (C,H,W)
        any = any || gmaps[sc];
    }
    if (!any) return 0;
    const int tiles = cdiv(M, DF_BM);
    double flops = 0;   // FLOPs issued (profile mode only; synchronises to read the scale-activity mask)
    if (srf_prof_on()) {
        std::vector<uint8_t> hm(tiles, 0x1f);
        if (hipMemcpyAsync(hm.data(), tile_mask, tiles, hipMemcpyDeviceToHost, s) == hipSuccess) (void)hipStreamSynchronize(s);
        for (int t = 0; t < tiles; ++t) {
            const int rows = M - t * DF_BM < DF_BM ? M - t * DF_BM : DF_BM;
            for (int sc = 0; sc < SCENERF_N_SCALES; ++sc)
                if (((hm[t] >> sc) & 1) && gmaps[sc]) flops += 2.0 * rows * (double)cfg->map_C[sc] * DF_K;
        }
    }
    SrfLaunchScope ps(s, w->d_out == 2 ? "gemm_dfeat_scatter/g" : "gemm_dfeat_scatter", flops, 0);
    // the finest level (every tile has it: 80 channels = 3 column tiles, one pass) and the coarser ones (a quarter of the tiles at
    // KITTI's geometry; 5 column tiles per pass) as two launches: a tile's dH rows are read once per launch that touches it
    const bool fine3 = gmaps[0] && cfg->map_C[0] <= 96;
    unsigned have = 0;
    for (int sc = 0; sc < SCENERF_N_SCALES; ++sc) have |= gmaps[sc] ? 1u << sc : 0u;
    if (fine3) {
        p.levels = 1u;
        p.live = p.levels & have;
        p.direct = 1;     // level 0 alone: <= 3 column tiles = one pass per tile, one workgroup per tile
        // ... except in the last round of workgroups (two per CU): 1,200 tiles are 2.34 rounds of 512 and every tile takes the same
        // ~50 us, so the 176 tiles of the third round are split along K over as many workgroups as fit it
        int cus1 = 256;
        (void)hipDeviceGetAttribute(&cus1, hipDeviceAttributeMultiprocessorCount, srf_device());
        const int slots = DF_WGS * cus1, full = tiles / slots * slots, tail = tiles - full;
        p.split_from = tiles; p.split_n = 1;
        if (tail > 0 && slots / tail >= 2) { p.split_from = full; p.split_n = slots / tail > 4 ? 4 : slots / tail; }
        dfeat_kernel<3><<<p.split_from + (tiles - p.split_from) * p.split_n, DF_THREADS, df_lds(3), s>>>(p);
        p.direct = 0;
    }
    p.levels = fine3 ? 30u : 31u;
    bool rest = false;
    for (int sc = 0; sc < SCENERF_N_SCALES; ++sc) rest = rest || (((p.levels >> sc) & 1u) && gmaps[sc]);
    p.live = p.levels & have;
    // (r04: this launch in FRONT of the finest level's -- it is the step's tail, 77 us alone -- changes nothing: whichever of the two runs
    // beside the batched weight gradients takes ~610 us there, the other ~150 us behind it: 2.496 / 2.497 against 2.485-2.496 ms)
    // the coarser levels: a quarter of the tiles at KITTI's geometry, so two workgroups per CU take the items in turn (a workgroup
    // that finds no item still pays the scan: 15 us per launch with one workgroup per tile and nothing to do)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, srf_device());
    const int grid2 = tiles < 2 * cus ? tiles : 2 * cus;
    if (rest) dfeat_kernel<DF_NTB><<<grid2, DF_THREADS, df_lds(DF_NTB), s>>>(p);
    SRF_LAUNCH_CHECK("dfeat_kernel");
    return 0;
}
