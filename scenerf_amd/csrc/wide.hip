// Fused ResnetFC kernels for gfx950 on 128-row blocks, ONE wave per SIMD (bf16 operands).  MODE 0: the whole 7-GEMM forward trunk
// (lin_in + lin_z.0, three residual blocks fc_0 / fc_1 + lin_z.b) and lin_out; MODE 1: the 6-GEMM dgrad chain of the blocks in the
// backward pass (for b = 2, 1, 0: dN_b = (dH_{b+1} W1_b) * [N_b > 0], dH_b = dH_{b+1} + (dN_b W0_b) * [H_b > 0]), with lin_out's input
// gradient dH3 = (d_logits W_out) * [H3 > 0] made in its prologue (MODE 2: the same chain on a dH3 tile read from memory).
// reference scenerf/models/resnetfc.py:41-57,133-164.
//
// Why this shape (measured history: DESIGN.md section 5).  A 64-row block needs all 16 KiB of a K chunk's weights per 256 MFMA cycles:
// 64 B/clk/CU, twice what a CU pulls from L2, so every 64-row design (fused.hip) saturates near 45 % MFMA.  A 128-row
// block halves the weight bytes per FLOP; its 128 x 512 fp32 accumulators are 256 KiB -- HALF of the CU's register file:
//   * 4 waves, one per SIMD, each with the whole 512-register budget: wave w owns output columns [128 w, 128 w + 128) for all 128 rows
//     = 4 x 4 MFMA 32x32x16 tiles = 256 accumulators held in a[0:255] (the accumulator file) by inline-asm MFMAs on literal registers;
//     the 256 architectural VGPRs hold the residual stream of the wave's tile (packed bf16, 128 registers: rounded exactly where the
//     other kernels round H_b), a 4-deep weight ring (64) and one set of activation fragments (16), re-loaded row tile by row tile;
//   * the hot loop -- the 32 chunks of a hidden layer, resident operand -- is branch-free: per chunk 16 MFMAs (512 cycles) against
//     4 LDS fragment reads, 4 weight loads (SGPR base + immediate offsets), 2 address instructions, one counted s_waitcnt; the
//     fragment address of chunk k is base ^ ((k & 7) << 5) + (k >> 3) * 256 (the A buffer's XOR swizzle only involves k mod 8);
//   * weights: the wave streams exactly ITS 4 KiB of a w_stream block (four coalesced 1-KiB global_load_dwordx4, already in fragment
//     layout) four chunks ahead -- every weight byte enters the CU once per 128 rows; nothing is shared between waves in a K loop:
//     no barrier in it.  The ring's first fill is plain loads; inside the K loop the loads are inline asm (saddr form: H_LW) and every
//     use sits behind a hand-counted `s_waitcnt vmcnt` (at the top of a chunk the 12 youngest loads -- three chunks -- plus the
//     stream-out's stores may be outstanding).  The compiler does NOT know that an asm load's destination is written later: the ring
//     registers must stay live until everything requested has landed -- the loop requests four chunks past the last one, so the
//     kernel's tail begins with `s_waitcnt vmcnt(0)` (round 6: without it a late load overwrote a store address of the tail, an
//     aperture violation in 1 of 12-46 processes on a busy GPU; tests/test_wide_isa.py guards the wait);
//   * the resident A operand (relu of the previous layer / the running gradient, 128 rows x 1 KiB, XOR-swizzled 16-byte slots) fills
//     128 KiB of LDS; the streamed operand of the lin_in / lin_z segments (X3 / Z columns, 4 KiB per chunk) goes global -> LDS by DMA
//     (global_load_lds) in ROUNDS: all of layer 0's chunks at once into the (still unused) A buffer + a 24-KiB stage, the lin_z tails of
//     layers 2 and 4 through the stage -- which at the common scale mask (finest level only: 5 chunks) still holds them from layer 0,
//     so those layers issue no DMA at all; other masks re-fetch per round of 6 chunks, the first round issued at the layer's start;
//   * layer epilogue: accumulators + bias (+ residual) -> bf16 -> relu -> A buffer, between two barriers (backward: gate by the
//     forward's sign bits instead of bias / relu); lin_out (512 -> 4) runs on the matrix cores too, from the resident H3 tile against a
//     three-term bf16 split of its fp32 weights; the finished layer is streamed out of the A buffer (coalesced 16-byte pieces +
//     sign bits, as in fused.hip) one piece per chunk of the next layer.
// Results: same rounding points as fused.hip; the forward adds the bias after the K sum instead of before it and sums
// layer 0's K segments in a different order, so activations agree with those kernels to the last bf16 ulp, not bit for bit; the
// backward chain has no bias and the same K order: bit-identical to fused.hip's and to the per-layer dgrad GEMMs (tests/test_gpu_stages.py).
#include "fused.h"
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_h;

#define H_BM 128
#define H_THREADS 256
#define H_D 4                                   // weight ring depth in chunks (= chunks per group of the unrolled loop)
#define H_ABUF (H_BM * F_AROW)                  // 131072
#define H_ZCAP 6                                // streamed-operand stage: chunks of 128 rows x 32 B
#define H_ZS H_ABUF
#define H_XB (H_ZS + H_ZCAP * 4096)             // 155648: 8 KiB of epilogue operands, fetched by DMA at the start of the layer's K loop --
                                                // forward: the layer's bias (2 KiB, two buffers), backward: the 128 rows' gate bits (8 KiB)
#define H_LDS (H_XB + 8192)                     // 163840: all of the CU's LDS
#define H_CAP0 (H_ZCAP + 32)                    // layer 0: stage + the A buffer (nothing resident yet)
// table set (F_MAXCH ints per tile mask): [0] chunks in total, [1] chunks of layer 0, [2] chunks of a lin_z tail (both incl. padding,
// multiples of H_D), [3] REAL chunks of a lin_z tail, [4 + c] descriptor of chunk c in execution order.  Descriptor bits as in fused.h,
// [25] = padding chunk (weights are loaded, MFMAs skipped).  Staged chunks (src != 0) appear in the order they are staged: list
// position = stage slot (mod the round capacity); layer 0 stages its Z chunks first so that they stay in the stage for layers 2 and 4.
#define HD_SKIP(d) (((d) >> 25) & 1)
#define H_HDR 4

// 16 bytes per lane, global -> LDS, no VGPR round trip: source = uniform base (SGPR pair) + 32-bit per-lane offset, destination = M0
// (wave-uniform LDS address) + 16 * lane.  Inline asm: invisible to the compiler's wait counting (every consumer waits by hand).
__device__ static inline void h_glds16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}

// ---- accumulators: a[0:255], tile (i, j) at a[16 (4 i + j) : +15]; MFMA = C^T tile (rows = outputs n, columns = activation rows m).
// Written and read BY NAME in inline asm: the kernel declares the whole accumulator file clobbered once, and must keep its own
// architectural-register pressure below 256 -- on gfx950 the register allocator otherwise moves live ranges into accumulator
// registers it believes free (tools/asmcheck.sh counts compiler-made v_accvgpr_* / scratch: both must be 0).  Handing the tiles to
// the compiler as "a"-constrained values instead was tried: it spills and copies them (and the in-flight weight ring) at loop joins.
typedef u32x4_h hfrag;   // one MFMA operand fragment: 8 bf16 = 4 registers
template <int B> __device__ __forceinline__ void h_mfma(const hfrag w, const hfrag a) {
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(a), "i"(B), "i"(B + 15));
}
template <int B> __device__ __forceinline__ void h_mfma0(const hfrag w, const hfrag a) {   // first chunk of a layer: C = 0
    asm volatile("v_mfma_f32_32x32x16_bf16 a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(a), "i"(B), "i"(B + 15));
}
template <int R> __device__ __forceinline__ float h_acc() {
    float v;
    asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "i"(R));
    return v;
}
// the four MFMAs of row tile I of a chunk: w[j] = weight fragments of the wave's four 32-column tiles, a = the row tile's activations
template <bool ZERO, int I> __device__ __forceinline__ void h_row(const hfrag (&w)[4], const hfrag a) {
    if (ZERO) { h_mfma0<16 * (4 * I)>(w[0], a); h_mfma0<16 * (4 * I + 1)>(w[1], a); h_mfma0<16 * (4 * I + 2)>(w[2], a); h_mfma0<16 * (4 * I + 3)>(w[3], a); }
    else { h_mfma<16 * (4 * I)>(w[0], a); h_mfma<16 * (4 * I + 1)>(w[1], a); h_mfma<16 * (4 * I + 2)>(w[2], a); h_mfma<16 * (4 * I + 3)>(w[3], a); }
}

// the wave's 4 KiB of a 16-KiB w_stream block: four 1-KiB loads, uniform base + per-lane offset + immediate.  Plain loads on purpose:
// the compiler counts vmcnt for them (the inline-asm stores / DMA in between only make its waits conservative -- loads return in
// order, so "at most N outstanding" still covers the load it waits for), and a register copy it may insert at a control-flow join
// waits for the data instead of copying a register whose load is still in flight.
typedef const __attribute__((address_space(1))) hfrag* glb_frag_p;
__device__ __forceinline__ void h_load_w(hfrag (&w)[4], const char* sbase, const unsigned voff) {
#ifdef H_VAR_NOW   // (development variants, tools/wide_cycles.py: where do the K loop's cycles go)
    return;
#endif
    const char* b = sbase + voff;
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = *(glb_frag_p)(uintptr_t)(b + 1024 * j);
}

__device__ static inline void h_store16(void* p, uint4 v) {
#ifndef H_VAR_ASMST
    // plain stores: the compiler counts them in vmcnt together with the weight loads (stores retire in issue order with the loads on
    // gfx9-class counters), so "wait for this chunk's weights" is exactly "all but the N youngest operations" -- with inline-asm stores
    // its N ignored up to eight younger stores and the wait reached into the next chunks' loads
#ifndef H_VAR_PLAINST
    // non-temporal: each activation is written once and read again only by a later kernel (measured: the kernels 3 % faster -- the next
    // block's staging reads no longer queue behind these lines -- at an unchanged step time)
    __builtin_nontemporal_store(v.x, (uint32_t*)p); __builtin_nontemporal_store(v.y, (uint32_t*)p + 1);
    __builtin_nontemporal_store(v.z, (uint32_t*)p + 2); __builtin_nontemporal_store(v.w, (uint32_t*)p + 3);
#else
    *(uint4*)p = v;
#endif
    return;
#endif
    const u32x4_h t = {v.x, v.y, v.z, v.w};
    // (s_nop: a 16-byte store reads its data registers after issue; the compiler does not know this statement is a store)
#ifdef H_VAR_SC1
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif defined(H_VAR_NT)
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif defined(H_VAR_SC01)
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#elif defined(H_VAR_SC1NT)
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#else
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
#endif
}
__device__ static inline void h_store1(void* p, uint32_t v) {
#ifndef H_VAR_ASMST
    *(uint8_t*)p = (uint8_t)v;
    return;
#endif
    asm volatile("global_store_byte %0, %1, off" ::"v"(p), "v"(v) : "memory");
}
// min of the two 16-bit halves with (1, 1): "is this bf16 non-zero" for a rectified pair, one instruction.  Inline asm: written as an
// elementwise min the compiler turns min(x, 1) into compare + select per half -- 30 vector instructions per stream-out piece for the 8
// sign bits instead of 9 (r03: that was most of what saving a layer's output cost the K loop, ~120 of 870 cycles per chunk).
__device__ static inline uint32_t h_pk_min_u16(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

// one quad (four consecutive outputs of one activation row) of the layer epilogue.
//   forward : v = acc + bias ; residual layers: h = bf16(h + v), out = relu(h) ; fc_0 layers: out = relu(bf16(v))
//   backward: v = gate ? acc : 0 ; dH layers: h = bf16(h + v), out = h ; dN layers: out = bf16(v)      (gate = 4 sign bits of the quad)
template <int MODE, bool is_res, int I, int J, int Q>
__device__ __forceinline__ void h_epi_quad(const float4 bias, const uint32_t gate, uint32_t (&hp)[8], char* wrow,
                                           const int slot0, const int axor) {
    // (the accumulator reads are volatile statements: program order = liveness, so a quad needs ~10 temporaries)
    constexpr int B = 16 * (4 * I + J) + 4 * Q;
    float v0 = h_acc<B>(), v1 = h_acc<B + 1>(), v2 = h_acc<B + 2>(), v3 = h_acc<B + 3>();
    if (MODE == 0) {
        v0 += bias.x; v1 += bias.y; v2 += bias.z; v3 += bias.w;
    } else {
        // gate bit -> all-ones / zero mask (one v_bfe_i32), value & mask: two instructions per value and no VCC round trip
        // (compare + select was three, plus the wait states of writing and reading VCC)
        auto gsel = [](const float v, const uint32_t g, const int bit) __attribute__((always_inline)) {
            uint32_t m;   // (inline asm: written in C, instcombine turns "and with a sign-extended bit" back into compare + select)
            asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(g), "n"(bit));
            return __uint_as_float(__float_as_uint(v) & m);
        };
        v0 = gsel(v0, gate, 8 * Q); v1 = gsel(v1, gate, 8 * Q + 1); v2 = gsel(v2, gate, 8 * Q + 2); v3 = gsel(v3, gate, 8 * Q + 3);
    }
    uint32_t p0, p1;
    if (is_res) {
        p0 = pack_bf16x2(v0 + bf16lo(hp[2 * Q]), v1 + bf16hi(hp[2 * Q]));
        p1 = pack_bf16x2(v2 + bf16lo(hp[2 * Q + 1]), v3 + bf16hi(hp[2 * Q + 1]));
        hp[2 * Q] = p0;
        hp[2 * Q + 1] = p1;
    } else {
        p0 = pack_bf16x2(v0, v1);
        p1 = pack_bf16x2(v2, v3);
    }
    uint2 o;   // outputs 32 j + 8 q + 4 hi + {0..3}: one 8-byte LDS write
    o.x = MODE == 0 ? relu_bf16x2(p0) : p0;
    o.y = MODE == 0 ? relu_bf16x2(p1) : p1;
    *(uint2*)(wrow + (((slot0 + Q) ^ axor) << 4)) = o;
}

#ifdef H_CYC   // development build (SRF_EXTRA_FLAGS=-DH_CYC): per-block time stamps of wave 0 (tools/wide_cycles.py)
__device__ unsigned long long* g_wide_cyc = nullptr;
extern "C" int scenerf_hip_test_wide_cyc(unsigned long long* ptr) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wide_cyc), &ptr, sizeof(ptr)); }
#define H_STAMP()                                                                                   \
    if (wvu == 0 && g_wide_cyc && ci < 64) {                                                        \
        if (lane == 0) g_wide_cyc[(size_t)blockIdx.x * 64 + ci] = __builtin_amdgcn_s_memtime();     \
        ++ci;                                                                                       \
    }
#else
#define H_STAMP()
#endif

// MODE_T: 0 / 3 = forward, 1 / 4 = backward chain with lin_out's prologue, 2 = backward chain on a staged dH3 tile.  0 and 1 are the
// instantiations for launches whose every block is full (M a multiple of 128) and whose every layer is saved (training): the stream-out's
// stores are then CERTAIN -- one 16-byte store (+ one sign-bit byte store, forward) per resident chunk -- and the K loop's waits count
// them (WSK, H_WAITW); 3 and 4 are the same kernels with the load-only count and the stores behind their row / pointer checks
// (inference, partial blocks); 2 (tests, A/B runs) starts with a run that streams nothing out and keeps the load-only count too.
template <int MODE_T>
__global__ __launch_bounds__(H_THREADS) void mlp_wide_kernel(FusedArgs p) {
    constexpr int MODE = MODE_T == 3 ? 0 : MODE_T == 4 ? 1 : MODE_T;
#ifdef H_VAR_WS0      // (development: the load-only count everywhere = the kernels as they were before round 5, for same-box A/B runs)
    constexpr int WSK = 0;
#else
    constexpr int WSK = MODE_T == 0 ? 2 : MODE_T == 1 ? 1 : 0;
#endif
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the accumulator file is this kernel's: a[0:255] are written by name in the MFMA statements
    asm volatile("" ::: "a0", "a15", "a16", "a31", "a32", "a63", "a64", "a95", "a96", "a127", "a128", "a159", "a160", "a191", "a192", "a223",
                 "a224", "a255");
    char* const Abuf = lds;
    if ((unsigned)(uintptr_t)lds & 255u) __builtin_trap();   // the fragment addresses XOR bits 5..7 of the absolute LDS address
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef H_CYC
    int ci = 0;
#endif
    H_STAMP()
    const int m0 = blockIdx.x * H_BM;
    const unsigned mask = MODE == 0 ? __builtin_amdgcn_readfirstlane((unsigned)p.tile_mask[m0 / SCENERF_TILE_ROWS] & 31u) : 0u;
    // this tile mask's table, read with scalar loads (constant address space) a group of four chunks at a time, two groups ahead
    const desc_ptr tab = (desc_ptr)(uintptr_t)(p.desc + mask * F_MAXCH);
    typedef int h_int4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) h_int4* desc4_ptr;
    const int n0 = tab[1];     // layer 0's staged chunks (forward), incl. padding
    const int nz = tab[2];     // a lin_z tail's staged chunks, incl. padding
    const int nzr = tab[3];    // ... real ones
    const bool zres = nzr <= H_ZCAP;                           // the stage keeps layer 0's Z chunks for layers 2 and 4
    // ---- L2 warm-up (round 6).  The first dispatch round puts one block on every CU at the same instant, on an L2 that the previous
    // kernels have filled with their own streams: every CU of an XCD then misses on the SAME weight block at the same time, chunk after
    // chunk, and the four-chunk ring cannot cover a trip to memory -- blocks of round 0 took 302k / 307k cycles (forward / dgrad chain)
    // against 236k / 192k for the later rounds, whose weights are L2 hits (tools/wide_cycles.py, profiles/r06_c_*).  So the blocks of
    // round 0 share the job of touching the launch's whole weight stream once, up front: block b sits on XCD b % 8 (dispatch order: a
    // speed assumption only), position (b >> 3) inside that XCD's round; position q touches chunk pairs q, q + npos, ... of this block's
    // own chunk table -- one 16-byte piece per 128-byte line, thread t line t & 127 of chunk 2 * pair + (wave >> 1).  The pieces go
    // global -> LDS by DMA (no destination register that a late return could clobber) into the second half of H_XB, which nothing has
    // written yet: whatever is staged there later is issued later and lands later (returns are in order), and every counted wait of the
    // K loops covers these oldest entries of the queue.
    // (launches of less than one full round -- the gaussian head's 38 blocks -- are slower with it, 103 against 98 us: each of their few
    // blocks would fetch a fifth of the stream before it starts)
    if (p.warm > 0 && gridDim.x >= 256 && blockIdx.x < 256) {
        const int npos = 32;
        const int nch = tab[0];
        const unsigned wdst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lds + H_XB + 4096 + wvu * 1024);
        for (int pr = (int)(blockIdx.x >> 3) % npos; 2 * pr < nch; pr += npos) {
            const int cw = 2 * pr + (wvu >> 1);
            if (cw < nch) h_glds16((const char*)p.Wst + ((size_t)FD_Z(tab[H_HDR + cw]) << 14), (unsigned)((tid & 127) << 7), wdst);
        }
    }
    // lin_out's bias, fetched now (scalar registers): a load at the tail would sit behind 128 KiB of stores in the in-order vmcnt queue
    float bo[4] = {0.f, 0.f, 0.f, 0.f};
    if (MODE == 0 && p.logits) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < p.d_out) bo[j] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(p.b_out[j])));
    }

    // ---- per-lane values.  Everything derived from the lane index is RE-derived from an opaque copy where it is needed: the 256
    // architectural registers are spoken for (residual stream 128, weight ring 64, fragments 16), and a compiler that hoists a few
    // dozen loop-invariant addresses has nowhere to put them (a[0:255] hold the accumulators; tools/asmcheck.sh refuses a build with
    // a compiler-made v_accvgpr_* or scratch).
    const char* const Wb = (const char*)p.Wst;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    int ln = lane;   // the opaque copy (refreshed by H_LANE)
#define H_LANE() asm volatile("" : "+v"(ln))
    typedef const __attribute__((address_space(3))) hfrag* lds_frag_p;
    // byte offset of this lane's 16 bytes inside a w_stream block ([512 rows n][32 B], halves swapped when (n >> 3) & 1; tile j: + 1 KiB)
    auto w_lane = [&]() __attribute__((always_inline)) { return (unsigned)(((wvu * 128 + (ln & 31)) * 2 + ((ln >> 5) ^ ((ln >> 3) & 1))) * 16); };
    // resident operand, k-half of this lane at column chunk 0: row (ln & 31) of row tile 0, slot hi ^ (row & 15)
    auto a_lane = [&]() __attribute__((always_inline)) { return lds0 + (unsigned)((ln & 31) * F_AROW + (((ln >> 5) ^ (ln & 15)) << 4)); };
    // staged chunk (128 rows x 32 B, halves swapped when (row >> 3) & 1): this lane's 16 bytes of row tile 0
    auto s_lane = [&]() __attribute__((always_inline)) { return (unsigned)((ln & 31) * 32 + (((ln >> 5) ^ ((ln >> 3) & 1)) << 4)); };
    // stage slot s of a round -> LDS offset: the 24-KiB stage first, then (layer 0 only) the A buffer
    auto slot_off = [&](const int s) __attribute__((always_inline)) { return s < H_ZCAP ? (unsigned)(H_ZS + s * 4096) : (unsigned)((s - H_ZCAP) * 4096); };

    // ---- layer output -> HBM: the A buffer of the finished layer is streamed out one 16-byte piece per thread per chunk of the NEXT
    // layer (32 pieces: rows 4 s .. 4 s + 3, one full row per wave), forward: rectified values + sign bits (fused.hip)
    // (wave-uniform state stays scalar: where the layer being streamed out goes -- fetched from the kernel arguments once per layer,
    // not per piece -- and the next piece).  A piece is read from LDS at the top of a chunk and stored behind the chunk's first four
    // MFMAs, so neither the LDS latency nor a scalar load sits in front of an MFMA.
    char* sv_base = nullptr;
    uint8_t* sg_base = nullptr;
    int sv_ld2 = 0;
    int save_i = 32;
    auto save_read = [&]() __attribute__((always_inline)) {
        H_LANE();   // (address arithmetic recomputed per piece, under the MFMAs, instead of being hoisted into live registers)
        const int row = (4 * save_i + wvu) & (H_BM - 1);
        return *(const uint4*)(Abuf + row * F_AROW + ((ln ^ (row & 15)) << 4));
    };
    auto save_write = [&](const uint4 v) __attribute__((always_inline)) {
        if (save_i < 32) {
            const int row = 4 * save_i + wvu, slot = ln;
#ifdef H_VAR_NOSAVE
            if (false) {
#else
            if (sv_base && m0 + row < p.M) {
#endif
                h_store16(sv_base + (size_t)(m0 + row) * sv_ld2 + slot * 16, v);
                if (MODE == 0 && sg_base) {
                    uint32_t u = h_pk_min_u16(v.x, 0x00010001u);
                    u |= h_pk_min_u16(v.y, 0x00010001u) << 2;
                    u |= h_pk_min_u16(v.z, 0x00010001u) << 4;
                    u |= h_pk_min_u16(v.w, 0x00010001u) << 6;
                    h_store1(sg_base + (size_t)(m0 + row) * 64 + slot, (u | (u >> 15)) & 0xffu);
                }
            }
            ++save_i;
        }
    };
    auto save_piece = [&]() __attribute__((always_inline)) { save_write(save_read()); };

    // ---- the residual stream (forward) / running gradient (backward) of the wave's 128 x 128 tile: packed bf16,
    // hp[i][j][k] = elements 2k, 2k+1 of tile (i, j)
    uint32_t hp[4][4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) hp[i][j][k] = 0u;

    // (RES: compile-time copy of the layer's kind -- residual layers h += ..., fc_0 / dN layers out = ... -- no branch per quad)
    auto epilogue = [&](auto RES, const int layer) __attribute__((always_inline)) {
        constexpr bool is_res = decltype(RES)::value;
        const FusedLayer& L = p.layer[layer];
        H_STAMP()   // K loop done
        while (save_i < 32) save_piece();
        H_LANE();
        const int hi = ln >> 5, axor = ln & 15;
        // forward: bias of the wave's tile j, outputs 8 q + 4 hi + {0..3}; backward: the 128 gate bits of row (32 i + lane & 31).  Both
        // come out of LDS (H_XB), where xb_dma() put them at the start of this layer's K loop: from global memory a bias quad cost an L2
        // round trip every (j, q) -- 3k of an epilogue's 7-9k cycles (r03: H_VAR_NOBIAS) -- and the gate rows one exposed at the top.
        typedef float f32x4_h __attribute__((ext_vector_type(4)));
        typedef const __attribute__((address_space(3))) f32x4_h* lds_f4_p;
        typedef const __attribute__((address_space(3))) u32x4_h* lds_u4_p;
        auto ld_f4 = [](const unsigned a) __attribute__((always_inline)) { const f32x4_h v = *(lds_f4_p)(uintptr_t)a; return make_float4(v.x, v.y, v.z, v.w); };
        const unsigned bb = lds0 + H_XB + (MODE == 0 ? (unsigned)((layer & 1) * 2048 + (wvu * 128 + 4 * hi) * 4) : (unsigned)((ln & 31) * 64 + wvu * 16));
        // (quads in (j, q, i) order: the four row tiles of a (j, q) share one float4 of bias -- one in use, the next one on its way)
        float4 bn = make_float4(0.f, 0.f, 0.f, 0.f);
        uint4 gt[4];
        // (MFMA results are visible to v_accvgpr_read only after the pipeline has drained: 16 passes)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // every wave has finished reading the A buffer for this layer -- and its K loop's counted waits
                                          // have retired the DMA pieces it issued at the loop's start: H_XB is complete
        H_STAMP()   // everyone arrived
        if (MODE == 0) {
#ifndef H_VAR_NOBIAS   // (development: what the bias reads cost the epilogue)
            bn = ld_f4(bb);
#endif
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u32x4_h v = *(lds_u4_p)(uintptr_t)(bb + i * 32 * 64);
                gt[i] = make_uint4(v.x, v.y, v.z, v.w);
            }
        }
        char* const wr0 = Abuf + (ln & 31) * F_AROW + 8 * hi;
#define H_GATE(I, J) (MODE != 0 ? ((J) == 0 ? gt[I].x : (J) == 1 ? gt[I].y : (J) == 2 ? gt[I].z : gt[I].w) >> (4 * hi) : 0u)
#ifdef H_VAR_NOBIAS
#define H_BIAS_ON false
#else
#define H_BIAS_ON true
#endif
#ifdef H_VAR_EPI4
#define H_EPI_MIDSB
#else
#define H_EPI_MIDSB __builtin_amdgcn_sched_barrier(0);
#endif
#define H_EPI_JQ(J, Q)                                                                                          \
    {                                                                                                           \
        const float4 bq = bn;                                                                                   \
        if (MODE == 0 && (J) * 4 + (Q) < 15 && H_BIAS_ON) bn = ld_f4(bb + ((((J) * 4 + (Q) + 1) >> 2) * 32 + (((Q) + 1) & 3) * 8) * 4); \
        /* quads in pairs between scheduling barriers: two independent dependency chains to interleave (a quad alone is one serial   \
           chain of ~18 instructions with hazard nops); no barrier at all = one huge basic block whose scheduler hoists every quad's     \
           residual unpacking to the top -- 256 more live registers */                                             \
        h_epi_quad<MODE, is_res, 0, J, Q>(bq, H_GATE(0, J), hp[0][J], wr0, wvu * 16 + (J) * 4, axor);           \
        h_epi_quad<MODE, is_res, 1, J, Q>(bq, H_GATE(1, J), hp[1][J], wr0 + 32 * F_AROW, wvu * 16 + (J) * 4, axor); \
        H_EPI_MIDSB                                                                                             \
        h_epi_quad<MODE, is_res, 2, J, Q>(bq, H_GATE(2, J), hp[2][J], wr0 + 64 * F_AROW, wvu * 16 + (J) * 4, axor); \
        h_epi_quad<MODE, is_res, 3, J, Q>(bq, H_GATE(3, J), hp[3][J], wr0 + 96 * F_AROW, wvu * 16 + (J) * 4, axor); \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    }
#define H_EPI_J(J) H_EPI_JQ(J, 0) H_EPI_JQ(J, 1) H_EPI_JQ(J, 2) H_EPI_JQ(J, 3)
        H_EPI_J(0) H_EPI_J(1) H_EPI_J(2) H_EPI_J(3)
#undef H_EPI_J
#undef H_EPI_JQ
#undef H_GATE
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // A buffer complete
        H_STAMP()   // epilogue done
        sv_base = (char*)L.save;
        sv_ld2 = L.save_ld * 2;
        sg_base = MODE == 0 ? L.sign : nullptr;
        save_i = 0;
    };

    // ---- epilogue operands of `layer` -> H_XB, issued at the start of the layer's K loop (the K loop's counted waits retire them: every
    // wave waits, chunk after chunk, for all but its ~16 youngest vector-memory operations)
    auto xb_dma = [&](const int layer) __attribute__((always_inline)) {
        H_LANE();
        const FusedLayer& L = p.layer[layer];
        if (MODE == 0) {
            if (wvu < 2)   // forward: 512 floats of bias = two 1-KiB pieces
                h_glds16((const char*)L.bias + wvu * 1024, (unsigned)(ln << 4),
                         __builtin_amdgcn_readfirstlane(lds0 + H_XB + (layer & 1) * 2048 + wvu * 1024));
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i)   // backward: 128 rows x 64 B of the forward's sign bits (rows past M: the buffer is padded)
                h_glds16((const char*)L.sign + (size_t)m0 * 64 + (wvu * 2 + i) * 1024, (unsigned)(ln << 4),
                         __builtin_amdgcn_readfirstlane(lds0 + H_XB + (wvu * 2 + i) * 1024));
        }
    };

    // ---- staged chunks: DMA of chunk descriptor d (X3 / Z columns FD_Y .. +15 of the block's 128 rows) into LDS offset `off`: this
    // wave's 32 rows = one 1-KiB piece; lane -> row lane / 2, physical 16-byte slot lane & 1, fetching the logical slot
    // physical ^ ((row >> 3) & 1) (swizzle on the source)
    // (the block's first row goes into the 64-bit uniform base, the lane offset is relative to it: as an absolute 32-bit offset it wrapped
    //  beyond 865,900 rows of Z -- every no_grad chunk of more than 1,691 rays at N = 512; round 6, see fused.hip)
    const char* const gX3 = (const char*)p.X3 + (size_t)m0 * (3 * SCENERF_D_XENC * 2);
    const char* const gZ = (const char*)p.Z + (size_t)m0 * (SCENERF_D_LATENT * 2);
    auto dma = [&](const int d, const unsigned off) __attribute__((always_inline)) {
        const int lr = min(m0 + 32 * wvu + (ln >> 1), p.M - 1) - m0;           // (rows past M: clamped, computed, dropped)
        const unsigned pls = (unsigned)(((ln & 1) ^ ((ln >> 4) & 1)) << 4);
        const bool x3 = FD_SRC(d) == 1;
        const unsigned voff = (unsigned)lr * (x3 ? 3u * SCENERF_D_XENC * 2u : SCENERF_D_LATENT * 2u) + pls;
        const char* sb = (x3 ? gX3 : gZ) + FD_Y(d) * 2;
        h_glds16(sb, voff, __builtin_amdgcn_readfirstlane(lds0 + off + wvu * 1024));
    };
    // DMA of the real chunks [k0, k1) of a staged list starting at table entry c0 (list position = slot)
    auto dma_round = [&](const int c0, const int k0, const int k1) __attribute__((always_inline)) {
        for (int k = k0; k < k1; ++k) {
            const int d = tab[H_HDR + c0 + k];
            if (!HD_SKIP(d)) dma(d, slot_off(k - k0));
        }
    };

    // ---- the weight stream: chunk c's block four chunks ahead of its MFMAs.  d0..d3 = descriptors of the current group of H_D chunks
    // (ring slots 0..3), e0..e3 = of the next group; `dnv` = the group after that, on its way from the LDS table.
    hfrag wr[H_D][4], af[4];
    int c = 0;   // chunks issued so far (table position of the current group)
    int d0, d1, d2, d3, e0, e1, e2, e3;
    h_int4 dnv;   // (scalar registers)
    {
        const h_int4 a = *(desc4_ptr)(tab + H_HDR), b = *(desc4_ptr)(tab + H_HDR + 4);
        d0 = a.x; d1 = a.y; d2 = a.z; d3 = a.w;
        e0 = b.x; e1 = b.y; e2 = b.z; e3 = b.w;
        dnv = *(desc4_ptr)(tab + H_HDR + 8);
    }
#ifdef H_VAR_W0      // (development: weights from a window of H_VAR_W0 blocks only: 2 = L1-resident, 16 = L2-resident)
#define H_WPTR(d) (Wb + (size_t)(FD_Z(d) & (H_VAR_W0 - 1)) * 16384)
#else
#define H_WPTR(d) (Wb + (size_t)FD_Z(d) * 16384)
#endif
    // top of a group: its descriptors become current, the next group's come out of `dnv`, the one after is requested
#define H_GROUP_TOP()                                                                                 \
    {                                                                                                 \
        d0 = e0; d1 = e1; d2 = e2; d3 = e3;                                                           \
        e0 = dnv.x; e1 = dnv.y; e2 = dnv.z; e3 = dnv.w;                                               \
        c += H_D;                                                                                     \
        dnv = *(desc4_ptr)(tab + H_HDR + c + 2 * H_D);                                                \
    }

    // ---- one RESIDENT chunk: ring slot S, chunk KK (0..7) of a group of eight whose fragment base is `ag` (= lane base + 256 per
    // eight chunks).  Row tile i's four MFMAs, then -- its fragment register being free once they are issued -- the same row tile's
    // fragment of the NEXT chunk (a single fragment set, each piece re-loaded 12 MFMAs = 384 cycles before its next use; at a
    // layer's last chunk that prefetch reads past the row, harmlessly), then the ring slot's next weights and one piece of the
    // previous layer's output on its way to HBM.
    // Everything that is not an MFMA is PINNED between two particular MFMAs (a scheduling barrier after every MFMA + filler group): one
    // wave per SIMD hides about five single-issue instructions behind a 32-cycle MFMA and none beyond that, so ~50 fillers per chunk
    // have to be spread over the 16 gaps, not bunched behind every fourth MFMA (measured: 680 -> cycles per chunk without loads).
    // The stream-out piece of chunk k of a resident run is piece k (a run has exactly 32 chunks and a layer 32 pieces); its stores sit
    // behind ONE not-taken scalar branch (EXEC-masking them instead was measured: every write to EXEC drains the MFMA pipe, +400 cycles
    // per chunk).
// (the ring loads of the resident loop are inline asm: the compiler's own count merges conservatively at the loop's back edge and
// asked for vmcnt(11) in front of a chunk's first MFMA -- two chunks of slack, not four.  By hand: behind load j of chunk c at least
// 15 - j vector-memory operations are younger -- the rest of its chunk and the next three chunks' loads; stores and DMA in between
// only add to that -- and loads retire in order, so vmcnt(15 - j) in front of the first MFMA that reads fragment j is safe.)
#ifdef H_VAR_CLD
#define H_WAITW(I, J, WS)
#else
// WS = vector-memory operations a chunk issues BESIDES its four ring loads (round 5).  The counter is shared: with the stream-out's
// stores in the loop (training forward: one 16-byte store + one sign-bit byte store per chunk; backward chain: one store) the operations
// younger than load j of chunk c number 15 - j + 3 WS, and vmcnt(15 - j) made the wave wait for ~1.5 chunks' worth of YOUNGER operations
// as well -- the ring's four chunks of slack shrank to about two and a half, and every store had to be acknowledged by L2 within two
// chunks instead of four (the same kernel without stores, inference, ran at 0.48 of peak against 0.34).  The exact count holds where the
// three chunks in front are known to have issued WS operations each: chunks 4.. of a resident run in the instantiations whose stores are
// certain (WSK); a run's first four chunks (what ran before them is the previous run's tail or an epilogue) keep the load-only count.
#define H_WAITW(I, J, WS) if ((I) == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(15 - (J) + 3 * (WS)));
#endif
#define H_M(I, J, S, ZERO)                                                          \
    H_WAITW(I, J, WS_)                                                              \
    if (ZERO) h_mfma0<16 * (4 * (I) + (J))>(wr[S][J], af[I]);                       \
    else h_mfma<16 * (4 * (I) + (J))>(wr[S][J], af[I]);
#define H_SB() __builtin_amdgcn_sched_barrier(0);
// (round 6: the chunk's address arithmetic is kept off the vector unit and out of 64 bits -- the weight loads and the stream-out's
// stores are the SGPR-base forms (uniform 64-bit base + 32-bit lane offset + immediate): one address register per lane instead of two,
// no v_lshl_add_u64 / s_mul per chunk; the stream-out's row pointer is a running scalar, its sign-bit rows are reached by the store's
// immediate offset within a group of eight chunks, the piece's LDS address is one XOR of a per-group register)
#ifdef H_VAR_NOW
#define H_LW(S, J, OFF)
#elif defined(H_VAR_HALF)
#define H_LW(S, J, OFF) if ((J) < 2) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(wr[S][J]) : "v"(wo_), "s"(Wb));
#else
#define H_LW(S, J, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(wr[S][J]) : "v"(wo_), "s"(Wb));
#endif
#define H_RES_CHUNK(S, KK, ZERO, DW) \
    { \
        const unsigned x0_ = (KK) == 7 ? ag + 256u : ag ^ (unsigned)((((KK) + 1) & 7) << 5);                                                           \
        const bool sok_ = WSK > 0 || (sv_run && (m0 + 4 * (8 * g + (KK)) + wvu < p.M));                                                                \
        H_M(0, 0, S, ZERO)                                                                                                                             \
        const u32x4_h svv_ = *(lds_u4r_p)(uintptr_t)((svl ^ (unsigned)(((KK) & 3) << 6)) + (unsigned)((KK) * 4 * F_AROW));                              \
        const uint4 sv_ = make_uint4(svv_.x, svv_.y, svv_.z, svv_.w);                                                                                  \
        H_SB()                                                                                                                                         \
        H_M(0, 1, S, ZERO)                                                                                                                             \
        H_SB()                                                                                                                                         \
        H_M(0, 2, S, ZERO)                                                                                                                             \
        H_SB()                                                                                                                                         \
        H_M(0, 3, S, ZERO)                                                                                                                             \
        af[0] = *(lds_frag_p)(uintptr_t)(x0_);                                                                                                         \
        H_SB()                                                                                                                                         \
        H_M(1, 0, S, ZERO)                                                                                                                             \
        if (sok_) h_store16(svp + l16, sv_);                                                                                                           \
        svp += svs;                                                                                                                                    \
        H_SB()                                                                                                                                         \
        H_M(1, 1, S, ZERO)                                                                                                                             \
        uint32_t su_ = 0; if (MODE == 0) { su_ = h_pk_min_u16(sv_.x, 0x00010001u); su_ |= h_pk_min_u16(sv_.y, 0x00010001u) << 2; }                     \
        H_SB()                                                                                                                                         \
        H_M(1, 2, S, ZERO)                                                                                                                             \
        if (MODE == 0) { su_ |= h_pk_min_u16(sv_.z, 0x00010001u) << 4; su_ |= h_pk_min_u16(sv_.w, 0x00010001u) << 6; }                                 \
        H_SB()                                                                                                                                         \
        H_M(1, 3, S, ZERO)                                                                                                                             \
        af[1] = *(lds_frag_p)(uintptr_t)(x0_ + 32768u);                                                                                                \
        H_SB()                                                                                                                                         \
        H_M(2, 0, S, ZERO)                                                                                                                             \
        if (MODE == 0 && sok_ && (WSK > 0 || sg_base)) h_store1(sgp + l1 + (KK) * 256, (su_ | (su_ >> 15)) & 0xffu);                                   \
        H_SB()                                                                                                                                         \
        H_M(2, 1, S, ZERO)                                                                                                                             \
        const unsigned wo_ = wl + ((unsigned)FD_Z(DW) << 14);                                                                                          \
        H_SB()                                                                                                                                         \
        H_M(2, 2, S, ZERO)                                                                                                                             \
        H_SB()                                                                                                                                         \
        H_M(2, 3, S, ZERO)                                                                                                                             \
        const unsigned x1_ = x0_ + 65536u; af[2] = *(lds_frag_p)(uintptr_t)(x1_);                                                                      \
        H_SB()                                                                                                                                         \
        H_M(3, 0, S, ZERO)                                                                                                                             \
        H_LW(S, 0, 0)                                                                                                                                  \
        H_SB()                                                                                                                                         \
        H_M(3, 1, S, ZERO)                                                                                                                             \
        H_LW(S, 1, 1024)                                                                                                                               \
        H_SB()                                                                                                                                         \
        H_M(3, 2, S, ZERO)                                                                                                                             \
        H_LW(S, 2, 2048)                                                                                                                               \
        H_SB()                                                                                                                                         \
        H_M(3, 3, S, ZERO)                                                                                                                             \
        H_LW(S, 3, 3072)                                                                                                                               \
        af[3] = *(lds_frag_p)(uintptr_t)(x1_ + 32768u);                                                                                                \
        H_SB()                                                                                                                                         \
    }
    // 32 resident chunks (a K = 512 operand in the A buffer): four groups of eight; the very first chunk starts the accumulators at 0
    // head_cons: the chunks in front of this run were not a resident run's (a staged run, the prologue): its first four chunks wait on the
    // load-only count.  Behind another resident run (the epilogue in between only adds younger operations) the exact count holds from chunk 0.
    auto resident_run = [&](const bool head_cons) __attribute__((always_inline)) {
        H_LANE();
        const unsigned wl = w_lane();
        unsigned ag = a_lane();
        // (a resident run starts with all 32 pieces of the previous layer's output still to stream out, or with none)
        const bool sv_run = save_i == 0 && sv_base != nullptr;
        save_i = 32;
        // the stream-out of chunk k: row 4 k + wvu of the block, one 16-byte slot per lane.  LDS address of the piece: row * 1 KiB + ((lane ^
        // (row & 15)) << 4), and row & 15 = (4 k & 12) | wvu: a per-group register XOR ((k & 3) << 6), + k * 4 KiB as the read's immediate;
        // HBM: a running uniform row pointer + (lane << 4); sign bits: a per-group uniform pointer + lane + 256 k as the store's immediate
        typedef const __attribute__((address_space(3))) u32x4_h* lds_u4r_p;
        char* svp = sv_base + (size_t)(m0 + wvu) * sv_ld2;
        const size_t svs = (size_t)4 * sv_ld2;
        uint8_t* sgp = sg_base + (size_t)(m0 + wvu) * 64;
        // fragments of chunk 0 (the A buffer was completed behind the epilogue's second barrier)
        af[0] = *(lds_frag_p)(uintptr_t)(ag);
        af[1] = *(lds_frag_p)(uintptr_t)(ag + 32768u);
        af[2] = *(lds_frag_p)(uintptr_t)(ag + 65536u);
        af[3] = *(lds_frag_p)(uintptr_t)(ag + 98304u);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int WS_ = WSK;
        // (a run's first four chunks: what ran before them is not this loop -- the load-only count, all four fragments at once, behind a
        // scalar branch.  Two whole copies of the chunks with different immediates were tried: the register allocator parks the joined
        // live ranges in accumulator registers.)
#define H_RUN_HEAD() if (WSK > 0 && g == 0 && head_cons) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
            H_LANE();
            // (per-lane addresses of the group's stream-out pieces, re-derived from the opaque lane copy: nothing lane-derived is carried
            // around the loop except the fragment base and the weight offset)
            const unsigned l16 = (unsigned)ln << 4, l1 = (unsigned)ln;
            const unsigned svl = lds0 + (unsigned)((32 * g + wvu) * F_AROW) + (unsigned)((ln ^ wvu) << 4);
            if (g == 0) {
                H_RUN_HEAD()
                H_RES_CHUNK(0, 0, true, e0)
            } else {
                H_RES_CHUNK(0, 0, false, e0)
            }
            H_RUN_HEAD()
            H_RES_CHUNK(1, 1, false, e1)
            H_RUN_HEAD()
            H_RES_CHUNK(2, 2, false, e2)
            H_RUN_HEAD()
            H_RES_CHUNK(3, 3, false, e3)
            H_GROUP_TOP()
            H_RES_CHUNK(0, 4, false, e0)
            H_RES_CHUNK(1, 5, false, e1)
            H_RES_CHUNK(2, 6, false, e2)
            H_RES_CHUNK(3, 7, false, e3)
            H_GROUP_TOP()
            ag += 256u;
            sgp += 32 * 64;
        }
#undef H_RUN_HEAD
    };

    // ---- STAGED chunks (forward only): `ns` list entries (multiple of H_D, padding at the end), `nreal` of them real, staged in rounds
    // of `cap` slots.  `pre` = round 0 is already on its way (issued at the layer's start) or resident (cap slots hold it since layer 0:
    // pre = 2: no wait needed either).  Each chunk: fragments of the NEXT list position are prefetched from its slot (clamped to
    // the round); a round switch reloads them.
#define H_STG_ROW(I, S, ZERO, SKIP, X)                                              \
    if (!(SKIP)) {                                                                  \
        if (ZERO) h_row<true, I>(wr[S], af[I]);                                     \
        else h_row<false, I>(wr[S], af[I]);                                         \
    }                                                                               \
    af[I] = *(lds_frag_p)(uintptr_t)((X) + (I) * 1024u);                            \
    __builtin_amdgcn_sched_barrier(0);
#define H_STG_CHUNK(S, DC, DW, ZERO)                                                \
    {                                                                               \
        if (k == rend && k < nreal) {                                               \
            /* round switch: everyone is done with the previous round's slots; stage the next ones; wait; publish */ \
            if (k > 0) {                                                            \
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                  \
                __builtin_amdgcn_s_barrier();                                       \
            }                                                                       \
            if (k > 0 || pre == 0) dma_round(c0, k, min(k + cap, nreal));           \
            if (k > 0 || pre != 2) {                                                \
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    \
                __builtin_amdgcn_s_barrier();                                       \
            }                                                                       \
            rbeg = k; rend = k + cap;                                               \
            if (first) { H_STAMP() }   /* layer 0's operand has landed */                \
            const unsigned x_ = lds0 + slot_off(0) + sl;                            \
            af[0] = *(lds_frag_p)(uintptr_t)(x_); af[1] = *(lds_frag_p)(uintptr_t)(x_ + 1024u);   \
            af[2] = *(lds_frag_p)(uintptr_t)(x_ + 2048u); af[3] = *(lds_frag_p)(uintptr_t)(x_ + 3072u); \
            __builtin_amdgcn_sched_barrier(0);                                      \
        }                                                                           \
        const bool skip_ = HD_SKIP(DC) != 0;                                        \
        const unsigned xn_ = lds0 + slot_off(min(k + 1 - rbeg, cap - 1)) + sl;      \
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   /* (this chunk's weights may come from the resident loop's asm loads) */ \
        H_STG_ROW(0, S, ZERO, skip_, xn_) H_STG_ROW(1, S, ZERO, skip_, xn_) H_STG_ROW(2, S, ZERO, skip_, xn_) H_STG_ROW(3, S, ZERO, skip_, xn_) \
        h_load_w(wr[S], H_WPTR(DW), wl);                                            \
        if (save_i < 32) save_piece();                                              \
        ++k;                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                          \
    }
    auto staged_run = [&](const int ns, const int nreal, const int cap, const int pre, const bool first) __attribute__((always_inline)) {
        H_LANE();
        const unsigned wl = w_lane();
        const unsigned sl = s_lane();
        const int c0 = c;
        int k = 0, rbeg = 0, rend = 0;
#pragma unroll 1
        for (int g = 0; g < ns; g += H_D) {
            if (g == 0 && first) {
                H_STG_CHUNK(0, d0, e0, true)
            } else {
                H_STG_CHUNK(0, d0, e0, false)
            }
            H_STG_CHUNK(1, d1, e1, false)
            H_STG_CHUNK(2, d2, e2, false)
            H_STG_CHUNK(3, d3, e3, false)
            H_GROUP_TOP()
        }
    };

    H_STAMP()   // setup done
    // ---- prologue: the weight ring (chunks 0 .. 3)
    {
        H_LANE();
        const unsigned wl = w_lane();
        h_load_w(wr[0], H_WPTR(d0), wl); h_load_w(wr[1], H_WPTR(d1), wl); h_load_w(wr[2], H_WPTR(d2), wl); h_load_w(wr[3], H_WPTR(d3), wl);
    }
    if (MODE == 0) {
        // (the two epilogue variants never meet at a join: the residual stream's 128 registers are rewritten in one of them only, and
        // the register allocator does not coalesce such a join -- it parks the overflow in accumulator registers)
        xb_dma(0);
        staged_run(n0, n0, H_CAP0, 0, true);
        epilogue(std::true_type(), 0);
#pragma unroll 1
        for (int b = 0; b < 3; ++b) {
            const bool tail = b < 2 && nz > 0;
            xb_dma(1 + 2 * b);
            resident_run(true);        // (behind layer 0's / the previous block's staged chunks -- or, without a lin_z tail, a resident run: rare)
            epilogue(std::false_type(), 1 + 2 * b);
            if (tail && !zres) dma_round(c + 32, 0, min(H_ZCAP, nzr));   // round 0 of this layer's lin_z tail: the stage is idle until then
            if (b == 2 && p.logits) {
                // lin_out's weights (fp32 [d_out][512]) -> the stage, which nobody reads any more: 1 KiB pieces.  Retired (in issue
                // order) long before the tail, which splits them into bf16 triples.
                H_LANE();
                for (int k = wvu; k < p.d_out * 2; k += 4)
                    h_glds16((const char*)p.w_out + k * 1024, (unsigned)(ln << 4), __builtin_amdgcn_readfirstlane(lds0 + H_ZS + k * 1024));
            }
            xb_dma(2 + 2 * b);
            resident_run(false);       // (behind fc_0's resident run)
            if (tail) staged_run(nz, nzr, H_ZCAP, zres ? 2 : 1, false);
            epilogue(std::true_type(), 2 + 2 * b);
        }
    } else {
      if constexpr (MODE == 1) {
        // ---- lin_out's input gradient, made here instead of read: dH3 = (d_logits W_out) * [H3 > 0] (resnetfc.py:160-163 backwards).
        // A K = 4 product per output: as its own kernel it was an HBM round trip of the whole [M][512] tile (linout_bwd: 97 us per
        // step, r03_i_step_trace), as fp32 FMAs here 1,100 vector instructions per lane (r02: net zero).  On the matrix cores it is three
        // chunks: both fp32 operands are split into three bf16 terms (x = hi + mid + lo to 24 bits: every term product is exact in the
        // fp32 accumulator) and chunk t multiplies term t of d_logits with all three terms of W_out -- K slots 0..3 = w_hi, 4..7 = w_mid,
        // 8..11 = w_lo, 12..15 = 0 -- i.e. the fp32 product to fp32 accumulation accuracy.  The tile then takes the dH layers' epilogue
        // (gate by H3's sign bits, round to bf16, A buffer + running gradient) and leaves for HBM under the first layer's K loop like any
        // other layer's output (the weight gradients of fc_1.2 read it there).
        H_LANE();
        xb_dma(6);      // H3's gate bits -> H_XB
        {
            const int hi = ln >> 5;
            hfrag wf[4], a3[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = wvu * 128 + 32 * j + (ln & 31);
                float r0 = p.w_out[col], r1 = p.d_out > 1 ? p.w_out[SCENERF_D_HIDDEN + col] : 0.f;
                float r2 = p.d_out > 2 ? p.w_out[2 * SCENERF_D_HIDDEN + col] : 0.f, r3 = p.d_out > 3 ? p.w_out[3 * SCENERF_D_HIDDEN + col] : 0.f;
                uint32_t t[3][2];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    t[sp][0] = pack_bf16x2(r0, r1);
                    t[sp][1] = pack_bf16x2(r2, r3);
                    r0 -= bf16lo(t[sp][0]); r1 -= bf16hi(t[sp][0]);
                    r2 -= bf16lo(t[sp][1]); r3 -= bf16hi(t[sp][1]);
                }
                wf[j] = hi ? hfrag{t[2][0], t[2][1], 0u, 0u} : hfrag{t[0][0], t[0][1], t[1][0], t[1][1]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* const dr = p.dlog + (size_t)min(m0 + 32 * i + (ln & 31), p.M - 1) * p.d_out;   // (rows past M: clamped, computed, dropped)
                float r0 = dr[0], r1 = p.d_out > 1 ? dr[1] : 0.f, r2 = p.d_out > 2 ? dr[2] : 0.f, r3 = p.d_out > 3 ? dr[3] : 0.f;
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const uint32_t ta = pack_bf16x2(r0, r1), tb = pack_bf16x2(r2, r3);
                    r0 -= bf16lo(ta); r1 -= bf16hi(ta);
                    r2 -= bf16lo(tb); r3 -= bf16hi(tb);
                    a3[sp][i] = hfrag{ta, tb, ta, tb};
                }
            }
            h_row<true, 0>(wf, a3[0][0]); h_row<true, 1>(wf, a3[0][1]); h_row<true, 2>(wf, a3[0][2]); h_row<true, 3>(wf, a3[0][3]);
            h_row<false, 0>(wf, a3[1][0]); h_row<false, 1>(wf, a3[1][1]); h_row<false, 2>(wf, a3[1][2]); h_row<false, 3>(wf, a3[1][3]);
            h_row<false, 0>(wf, a3[2][0]); h_row<false, 1>(wf, a3[2][1]); h_row<false, 2>(wf, a3[2][2]); h_row<false, 3>(wf, a3[2][3]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the gate bits (no K loop in front of this epilogue to retire the DMA)
        epilogue(std::true_type(), 6);                     // running gradient = 0 + gated tile
        H_STAMP()   // incoming gradient tile made
      } else {
        // MODE 2 (tests, A/B runs): the incoming gradient tile dH3 as linout_bwd_kernel wrote it -> resident A buffer: one row (1 KiB) per
        // piece, 32 rows per wave; lane = physical slot, fetching the logical slot physical ^ (row & 15)
        H_LANE();
        for (int r = 0; r < 32; ++r) {
            const int row = 32 * wvu + r;
            const int gm = min(m0 + row, p.M - 1);
            h_glds16((const char*)p.dH3 + (size_t)gm * (p.dH_ld * 2), (unsigned)((ln ^ (row & 15)) << 4),
                     __builtin_amdgcn_readfirstlane(lds0 + row * F_AROW));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {   // the running gradient of the wave's tile, in the epilogue's layout
            const int hi = ln >> 5, axor = ln & 15;
            const char* const rd0 = Abuf + (ln & 31) * F_AROW + 8 * hi;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const uint2 v = *(const uint2*)(rd0 + i * 32 * F_AROW + (((wvu * 16 + j * 4 + q) ^ axor) << 4));
                        hp[i][j][2 * q] = v.x;
                        hp[i][j][2 * q + 1] = v.y;
                    }
        }
        H_STAMP()   // incoming gradient tile staged
      }
#pragma unroll 1
        for (int l = 0; l < 6; l += 2) {
            xb_dma(l);
            resident_run(l == 0);                 // (the first run follows the prologue)
            epilogue(std::false_type(), l);       // dN_b = ...
            xb_dma(l + 1);
            resident_run(false);
            epilogue(std::true_type(), l + 1);    // dH_b = dH_{b+1} + ...
        }
    }
#undef H_STG_CHUNK
#undef H_STG_ROW
#undef H_RES_CHUNK
#undef H_M
#undef H_SB
#undef H_LW
#undef H_GROUP_TOP
#undef H_WPTR
    // The weight ring's registers die here with loads still in flight: the branch-free K loop requests the blocks of the four chunks
    // BEHIND the last one (zero-padded descriptors: block 0) and nobody ever waits for them.  Those requests are inline asm (H_LW): the
    // compiler does not know that sixteen writes to wr[][] are pending, takes the registers for the tail's store addresses -- and a load
    // that lands late overwrites a pointer.  Found in round 6 as HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in 1 of 12-46 processes on
    // boxes whose GPU was busy with other work too, and in 2 of 12 processes when three ran at once (memory latency decides whether the
    // load or the compiler's reuse comes first; tools/preempt_stress.sh).  Everything this wave requested has landed before the
    // registers are handed back:
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H_LANE();
    // ---- tail: the last layer's output goes out four pieces at a time (four LDS reads in flight, then four stores)
    const bool do_out = MODE == 0 && p.logits;
    while (save_i < 32) {
        uint4 pv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (4 * (save_i + r) + wvu) & (H_BM - 1);
            pv[r] = *(const uint4*)(Abuf + row * F_AROW + ((ln ^ (row & 15)) << 4));
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) save_write(pv[r]);
    }
    H_STAMP()   // last layer streamed out
    if (do_out) {
        // lin_out on the rectified H3 tile still resident in the A buffer, on the matrix cores: logits^T = [w_hi ; w_mid ; w_lo] (12 of an
        // MFMA's 32 rows) x H3^T, the fp32 weights split into three bf16 terms (w = hi + mid + lo to 24 bits; the bf16 activations times
        // each term are exact in the fp32 accumulator), one MFMA per 16-wide K chunk and row tile -- a wave takes one row tile: 32 MFMAs.
        // (As fp32 FMAs on the vector unit this tail took 14k cycles per block: 1,024 dependent FMAs per lane, LDS round trips, a butterfly.)
        // Stage: raw w_out at H_ZS (8 KiB, DMA'd while the last layer ran), the 13 operand rows (4 hi, 4 mid, 4 lo, 1 zero) at
        // H_ZS + 8 KiB in the A buffer's layout (1 KiB rows, 16-byte slot s at s ^ (row & 15)).
        char* const wsp = lds + H_ZS + 8192;
        {
            const int t = wvu * 64 + ln, j = t >> 6, k8 = t & 63;   // thread: 8 consecutive k of weight row j
            const float* const raw = (const float*)(lds + H_ZS) + j * SCENERF_D_HIDDEN + k8 * 8;
            float w[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (j < p.d_out) {
                const float4 lo = *(const float4*)raw, hi4 = *(const float4*)(raw + 4);
                w[0] = lo.x; w[1] = lo.y; w[2] = lo.z; w[3] = lo.w; w[4] = hi4.x; w[5] = hi4.y; w[6] = hi4.z; w[7] = hi4.w;
            }
            uint32_t part[3][4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float r0 = w[2 * e], r1 = w[2 * e + 1];
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) {
                    const uint32_t pk = pack_bf16x2(r0, r1);
                    part[sp][e] = pk;
                    r0 -= bf16lo(pk);
                    r1 -= bf16hi(pk);
                }
            }
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const int row = 4 * sp + j;
                *(uint4*)(wsp + row * F_AROW + ((k8 ^ (row & 15)) << 4)) = make_uint4(part[sp][0], part[sp][1], part[sp][2], part[sp][3]);
            }
            if (t < 64) *(uint4*)(wsp + 12 * F_AROW + ((t ^ 12) << 4)) = make_uint4(0u, 0u, 0u, 0u);   // the zero row every other output row reads
        }
        __syncthreads();
        {
            const int n = min(ln & 31, 12), hi = ln >> 5;
            const unsigned wb = lds0 + H_ZS + 8192 + (unsigned)(n * F_AROW);
            const unsigned ab = lds0 + (unsigned)((32 * wvu + (ln & 31)) * F_AROW);
            const int ax = ln & 15, wx = n & 15;
            hfrag wf = *(lds_frag_p)(uintptr_t)(wb + ((hi ^ wx) << 4));
            hfrag xf = *(lds_frag_p)(uintptr_t)(ab + ((hi ^ ax) << 4));
            h_mfma0<0>(wf, xf);
#pragma unroll 4
            for (int cc = 1; cc < SCENERF_D_HIDDEN / F_BK; ++cc) {
                wf = *(lds_frag_p)(uintptr_t)(wb + (((2 * cc + hi) ^ wx) << 4));
                xf = *(lds_frag_p)(uintptr_t)(ab + (((2 * cc + hi) ^ ax) << 4));
                h_mfma<0>(wf, xf);
            }
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");   // (MFMA results -> v_accvgpr_read)
            // C^T tile: lane = activation row (ln & 31), output rows 8 (r >> 2) + (r & 3) + 4 hi: hi = 0 holds the hi term (r = j) and the lo
            // term (r = 4 + j), hi = 1 the mid term (r = j)
            const float a0[4] = {h_acc<0>(), h_acc<1>(), h_acc<2>(), h_acc<3>()};
            const float a1[4] = {h_acc<4>(), h_acc<5>(), h_acc<6>(), h_acc<7>()};
            const int row = 32 * wvu + (ln & 31);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float mid = __shfl_xor(a0[j], 32, 64);
                if (hi == 0 && j < p.d_out && m0 + row < p.M) p.logits[(size_t)(m0 + row) * p.d_out + j] = (a0[j] + mid) + a1[j] + bo[j];
            }
        }
    }
    H_STAMP()   // end
}

// ---- host: tables for the 32 scale masks of the forward (sets 0..31) and the backward chain (set 32)
int wide_table_build(const scenerf_cfg* cfg, std::vector<int>& tab) {
    tab.assign((size_t)33 * F_MAXCH, 0);
    int seg_off[5], off = 0;
    for (int i = 0; i < 5; ++i) { seg_off[i] = off; off += cfg->map_C[i]; }
    SRF_CHECK(off == SCENERF_D_LATENT, "wide mlp: map channels do not add up to the latent width");
    const int layer_k[7] = {3 * SCENERF_D_XENC + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN,
                            SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN};
    int layer_block0[7], nb = 0;
    for (int i = 0; i < 7; ++i) { layer_block0[i] = nb; nb += layer_k[i] / F_BK; }
    SRF_CHECK(nb < 1024, "wide mlp: w_stream block index does not fit the descriptor");
    for (int mask = 0; mask < 32; ++mask) {
        int* const hdr = tab.data() + (size_t)mask * F_MAXCH;
        int* ch = hdr + H_HDR;
        int n = 0;
        bool ok = true;
        auto seg = [&](int layer, int src, int a0, int w0, int len) {
            if (len % F_BK || a0 % F_BK || w0 % F_BK) ok = false;
            for (int k = 0; k + F_BK <= len; k += F_BK) {
                if (n >= F_MAXCH - H_HDR - 16) { ok = false; return; }
                ch[n++] = (layer_block0[layer] + (w0 + k) / F_BK) | (((a0 + k) / F_BK) << 10) | (src << 18) | (layer << 20);
            }
        };
        auto zsegs = [&](int layer, int wbase) {
            for (int i = 0; i < 5; ++i) {
                if ((mask >> i) & 1) seg(layer, 2, seg_off[i], wbase, cfg->map_C[i]);
                wbase += cfg->map_C[i];
            }
        };
        auto pad = [&](int layer) {   // padding chunks up to a multiple of H_D: resident source, block 0, MFMAs skipped
            while (n % H_D) ch[n++] = (layer << 20) | (1 << 25);
        };
        // layer 0: the Z chunks first (stage slots 0.., where layers 2 and 4 find them again), then [x_hi | x_lo | x_hi]
        zsegs(0, 3 * SCENERF_D_XENC);
        const int nzr = n;
        seg(0, 1, 0, 0, 3 * SCENERF_D_XENC);
        pad(0);
        hdr[1] = n;
        for (int b = 0; b < 3; ++b) {
            seg(1 + 2 * b, 0, 0, 0, SCENERF_D_HIDDEN);
            seg(2 + 2 * b, 0, 0, 0, SCENERF_D_HIDDEN);
            if (b < 2) {
                const int t0 = n;
                zsegs(2 + 2 * b, SCENERF_D_HIDDEN);
                pad(2 + 2 * b);
                hdr[2] = n - t0;
            }
        }
        SRF_CHECK(ok && n % H_D == 0, "wide mlp: segment lengths must be multiples of 16 and fit the descriptor table");
        for (int i = 0; i < n; ++i) {
            if (i + 1 == n || FD_LAYER(ch[i + 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 23;
            if (i == 0 || FD_LAYER(ch[i - 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 24;
        }
        hdr[0] = n;   // entries past n stay zero: prefetches past the end read block 0 and are never used
        hdr[3] = nzr;
    }
    {   // backward chain: six layers of 32 resident chunks, weight blocks after the forward ones
        int* const hdr = tab.data() + (size_t)32 * F_MAXCH;
        int* ch = hdr + H_HDR;
        int n = 0;
        for (int l = 0; l < 6; ++l)
            for (int i = 0; i < SCENERF_D_HIDDEN / F_BK; ++i, ++n)
                ch[n] = (nb + l * (SCENERF_D_HIDDEN / F_BK) + i) | (i << 10) | (l << 20) | (i + 1 == SCENERF_D_HIDDEN / F_BK ? 1 << 23 : 0) |
                        (i == 0 ? 1 << 24 : 0);
        hdr[0] = n;
        SRF_CHECK(nb + 6 * (SCENERF_D_HIDDEN / F_BK) == SCENERF_W_STREAM_BLOCKS, "wide mlp: w_stream block count");
    }
    return 0;
}

static SrfDescCache g_wide_table;

static int wide_attrs() {
    SRF_ONCE_PER_DEVICE(
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_wide_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_wide_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_wide_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS)));
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_wide_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_wide_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS));
    return 0;
}

int wide_prepare(const scenerf_cfg* cfg, hipStream_t s) {
    if (int e = wide_attrs()) return e;
    const int* d = nullptr;
    return srf_desc_cache_get(g_wide_table, cfg, s, wide_table_build, &d);
}

int launch_mlp_fwd_wide(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                        const scenerf_mlp_acts* a, hipStream_t s) {
    if (int e = wide_attrs()) return e;
    FusedArgs p = {};
    const int H = SCENERF_D_HIDDEN;
    const size_t sign_layer = (size_t)cdiv(M, SCENERF_TILE_ROWS) * SCENERF_TILE_ROWS * 64;
    auto sign = [&](int l) { return a->sign_bits ? a->sign_bits + l * sign_layer : nullptr; };
    p.layer[0] = {w->b_h[0], a->H[0], sign(0), 0, H};
    for (int b = 0; b < 3; ++b) {
        p.layer[1 + 2 * b] = {w->b_fc0[b], a->Nn[b], sign(1 + 2 * b), 1, H};
        p.layer[2 + 2 * b] = {w->b_h[b + 1], a->H[b + 1], sign(2 + 2 * b), 2, H};   // (H3's bits gate lin_out's input gradient: MODE 1's prologue)
    }
    p.Wst = w->w_stream;
    p.X3 = a->h0pre;
    p.Z = Z;
    p.tile_mask = tile_mask;
    if (int e = srf_desc_cache_get(g_wide_table, cfg, s, wide_table_build, &p.desc)) return e;
    p.M = M;
    p.w_out = w->w_out;
    p.b_out = w->b_out;
    p.logits = a->logits;
    p.d_out = w->d_out;
    p.warm = srf_warm_wide();
    double flops = 0;   // FLOPs actually issued (profile mode only; synchronises to read the scale-activity mask)
    if (srf_prof_on()) {
        const int tiles = cdiv(M, SCENERF_TILE_ROWS);
        std::vector<uint8_t> hm(tiles, 0x1f);
        if (hipMemcpyAsync(hm.data(), tile_mask, tiles, hipMemcpyDeviceToHost, s) == hipSuccess) (void)hipStreamSynchronize(s);
        for (int t = 0; t < tiles; ++t) {
            const int rows = M - t * SCENERF_TILE_ROWS < SCENERF_TILE_ROWS ? M - t * SCENERF_TILE_ROWS : SCENERF_TILE_ROWS;
            double kz = 0;
            for (int i = 0; i < 5; ++i)
                if ((hm[t] >> i) & 1) kz += cfg->map_C[i];
            flops += 2.0 * rows * 512.0 * (3.0 * SCENERF_D_XENC + 6.0 * SCENERF_D_HIDDEN + 3.0 * kz);
        }
    }
    SrfLaunchScope ps(s, w->d_out == 2 ? "mlp_fwd_fused/g" : "mlp_fwd_fused", flops, 0);
    // every block full and every layer + its sign bits saved: the instantiation whose K loop counts the stream-out's stores (kernel header)
    bool certain = M % H_BM == 0 && a->sign_bits != nullptr;
    for (int l = 0; l < 7; ++l) certain = certain && p.layer[l].save != nullptr;
    if (certain) mlp_wide_kernel<0><<<cdiv(M, H_BM), H_THREADS, H_LDS, s>>>(p);
    else mlp_wide_kernel<3><<<cdiv(M, H_BM), H_THREADS, H_LDS, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_wide_kernel<fwd>");
    return 0;
}

// Backward dgrad chain on the same kernel shape (results as launch_mlp_bwd_fused, fused.hip).  d_logits != NULL: lin_out's input gradient
// is made in the kernel's prologue from d_logits, w_out and H3's sign bits and written to dH's column block 3 by the kernel itself (the
// caller then only needs lin_out's WEIGHT gradients from linout_bwd); NULL: column block 3 of dH is read as the caller left it.
int launch_mlp_bwd_wide(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, int M, const scenerf_mlp_acts* a, void* dH, void* dN,
                        const float* d_logits, hipStream_t s) {
    if (int e = wide_attrs()) return e;
    FusedArgs p = {};
    const int H = SCENERF_D_HIDDEN;
    const size_t sign_layer = (size_t)cdiv(M, SCENERF_TILE_ROWS) * SCENERF_TILE_ROWS * 64;
    SRF_CHECK(a->sign_bits, "wide backward: the sign bits of the fused forward are missing");
    for (int b = 2; b >= 0; --b) {
        const int l = 2 * (2 - b);
        // forward layer order of the sign bits: H0, N0, H1, N1, H2, N2, H3
        p.layer[l] = {nullptr, (char*)dN + (size_t)b * M * H * 2, a->sign_bits + (size_t)(2 * b + 1) * sign_layer, 1, H};   // dN_b = (dH_{b+1} W1_b) * [N_b > 0]
        p.layer[l + 1] = {nullptr, (char*)dH + (size_t)b * H * 2, a->sign_bits + (size_t)(2 * b) * sign_layer, 2, 4 * H};   // dH_b = dH_{b+1} + (dN_b W0_b) * [H_b > 0]
    }
    p.layer[6] = {nullptr, (char*)dH + (size_t)3 * H * 2, a->sign_bits + (size_t)6 * sign_layer, 2, 4 * H};                 // dH3 = (d_logits W_out) * [H3 > 0]
    p.Wst = w->w_stream;
    p.dH3 = (const char*)dH + (size_t)3 * H * 2;
    p.dH_ld = 4 * H;
    p.dlog = d_logits;
    p.w_out = w->w_out;
    p.d_out = w->d_out;
    const int* desc = nullptr;
    if (int e = srf_desc_cache_get(g_wide_table, cfg, s, wide_table_build, &desc)) return e;
    p.desc = desc + 32 * F_MAXCH;
    p.M = M;
    p.warm = srf_warm_wide();
    SrfLaunchScope ps(s, w->d_out == 2 ? "mlp_bwd_fused/g" : "mlp_bwd_fused", 2.0 * M * 512.0 * (6.0 * 512.0 + (d_logits ? 48.0 : 0.0)), 0);
    if (d_logits && M % H_BM == 0) mlp_wide_kernel<1><<<cdiv(M, H_BM), H_THREADS, H_LDS, s>>>(p);    // (full blocks: the stores are certain)
    else if (d_logits) mlp_wide_kernel<4><<<cdiv(M, H_BM), H_THREADS, H_LDS, s>>>(p);
    else mlp_wide_kernel<2><<<cdiv(M, H_BM), H_THREADS, H_LDS, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_wide_kernel<bwd>");
    return 0;
}
