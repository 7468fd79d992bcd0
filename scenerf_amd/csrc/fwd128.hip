// Fused ResnetFC forward for gfx950 (bf16 operands), 128-row blocks, ONE wave per SIMD: the whole 7-GEMM trunk (lin_in + lin_z.0, three
// residual blocks fc_0 / fc_1 + lin_z.b) and lin_out in one kernel.  reference scenerf/models/resnetfc.py:133-164.
//
// Why this shape (measured history: DESIGN.md section 5).  A 64-row block needs all 16 KiB of a K chunk's weights per 256 MFMA cycles:
// 64 B/clk/CU, twice what a CU pulls from L2, so every 64-row design (fused.hip, stream.hip) saturates near 45 % MFMA.  A 128-row
// block halves the weight bytes per FLOP; its 128 x 512 fp32 accumulators are 256 KiB -- HALF of the CU's register file.  With 8 waves
// (2 per SIMD, 256 registers each) nothing else fits, and the two waves of a SIMD drift apart (issue arbitration favours the older
// one), which is what the round-1 128-row experiment measured.  So:
//   * 4 waves, one per SIMD, each with the whole 512-register budget: wave w owns output columns [128 w, 128 w + 128) for all 128 rows
//     = 4 x 4 MFMA 32x32x16 tiles = 256 accumulators, held in a[0:255] (the accumulator file) by inline-asm MFMAs on literal registers;
//     the 256 architectural VGPRs hold the residual stream of the wave's tile (packed bf16, 128 registers: rounded exactly where the
//     other kernels round H_b), a 4-deep weight ring (64) and one set of activation fragments (16), re-loaded row tile by row tile;
//   * 16 MFMAs (512 cycles) per chunk against 4 LDS reads + 4 weight loads: ~1 non-MFMA instruction per MFMA (the 64-row kernels issue
//     8-10), and no second wave to fight with;
//   * weights: the wave streams exactly ITS 4 KiB of a w_stream block (four coalesced 1-KiB global_load_dwordx4, already in fragment
//     layout) -- every weight byte enters the CU once per 128 rows; nothing is shared between waves in a K loop: no barrier in it;
//   * the resident A operand (relu of the previous layer, 128 rows x 1 KiB, XOR-swizzled 16-byte slots) fills 128 KiB of LDS; the
//     streamed operand of the lin_in / lin_z segments (X3 / Z rows, 4 KiB per chunk) goes global -> LDS by DMA (global_load_lds, each
//     wave its 32 rows) two chunks ahead into three 4-KiB stages, so that EVERY activation fragment is an LDS read: one load kind, no
//     branch that merges in-flight load results (a first version with three code paths -- LDS / X3 / Z -- made the compiler copy
//     loaded registers at the joins, each copy behind s_waitcnt vmcnt(0): the weight ring drained every chunk, 2,350 cycles per chunk
//     instead of 512).  Only streamed chunks (24 of 204 at scale mask 1) cost a barrier;
//   * layer epilogue: accumulators + bias (+ residual) -> bf16 -> relu -> A buffer, between two barriers; the finished layer is streamed
//     out of the A buffer (coalesced 16-byte pieces + sign bits, as in fused.hip) during the next layer's chunks.
// Results: same rounding points as fused.hip / stream.hip; the bias is added after the K sum instead of before it, so activations
// agree with those kernels to the last bf16 ulp, not bit for bit (tests/test_gpu_stages.py).
#include "fused.h"
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_h;

#define H_BM 128
#define H_THREADS 256
#define H_D 4                                   // weight ring depth in chunks (= chunks per group of the unrolled loop)
#define H_ABUF (H_BM * F_AROW)                  // 131072
#define H_BIAS H_ABUF                           // 7 layers x 2 KiB
#define H_NSTG 3                                // streamed-operand stages: 128 rows x 32 B each
#define H_STG (H_BIAS + 7 * 2048)               // (w_out, <= 8 KiB, takes the stages' place after the last layer)
#define H_TAB (H_STG + H_NSTG * 4096)           // this tile mask's chunk descriptors (+ read slack)
#define H_LDS (H_TAB + (F_MAXCH + 16) * 4)      // 160576 of 163840
// descriptor bits as in fused.h, except [25] = no-op chunk (padding a layer to a multiple of H_D chunks: loads happen, MFMAs do not)
// and [26:27] = stage of a streamed chunk (chunk index mod 3)
#define HD_SKIP(d) (((d) >> 25) & 1)
#define HD_STAGE(d) (((d) >> 26) & 3)

// 16 bytes per lane, global -> LDS, no VGPR round trip: source = uniform base (SGPR pair) + 32-bit per-lane offset, destination = M0
// (wave-uniform LDS address) + 16 * lane.  Inline asm: invisible to the compiler's wait counting (every consumer waits by hand).
__device__ static inline void h_glds16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}

// ---- accumulators: a[0:255], tile (i, j) at a[16 (4 i + j) : +15]; MFMA = C^T tile (rows = outputs n, columns = activation rows m)
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

__device__ static inline void h_store16(void* p, uint4 v) {
    const u32x4_h t = {v.x, v.y, v.z, v.w};
    // (s_nop: a 16-byte store reads its data registers after issue; the compiler does not know this statement is a store)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ static inline void h_store1(void* p, uint32_t v) { asm volatile("global_store_byte %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
typedef unsigned short h_ushort2 __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t h_pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(h_ushort2, a), __builtin_bit_cast(h_ushort2, b)));
}

// one (i, j) tile of the layer epilogue: accumulators + bias (+ residual) -> bf16 -> (residual) -> relu -> A buffer.  Four outputs at a
// time (the accumulator reads are volatile statements: program order = liveness), so a tile needs ~10 temporaries, not 40
template <int I, int J, int Q>
__device__ __forceinline__ void h_epi_quad(const float (&bias)[16], uint32_t (&hp)[8], const bool is_res, char* wrow, const int slot0, const int axor) {
    constexpr int B = 16 * (4 * I + J) + 4 * Q;
    const float v0 = h_acc<B>() + bias[4 * Q], v1 = h_acc<B + 1>() + bias[4 * Q + 1];
    const float v2 = h_acc<B + 2>() + bias[4 * Q + 2], v3 = h_acc<B + 3>() + bias[4 * Q + 3];
    uint32_t p0, p1;
    if (is_res) {   // wave-uniform
        p0 = pack_bf16x2(v0 + bf16lo(hp[2 * Q]), v1 + bf16hi(hp[2 * Q]));
        p1 = pack_bf16x2(v2 + bf16lo(hp[2 * Q + 1]), v3 + bf16hi(hp[2 * Q + 1]));
        hp[2 * Q] = p0;
        hp[2 * Q + 1] = p1;
    } else {
        p0 = pack_bf16x2(v0, v1);
        p1 = pack_bf16x2(v2, v3);
    }
    uint2 o;   // outputs 32 j + 8 q + 4 hi + {0..3}: one 8-byte LDS write
    o.x = relu_bf16x2(p0);
    o.y = relu_bf16x2(p1);
    *(uint2*)(wrow + (((slot0 + Q) ^ axor) << 4)) = o;
}
template <int I, int J>
__device__ __forceinline__ void h_epi_tile(const float (&bias)[16], uint32_t (&hp)[8], const bool is_res, char* wrow, const int slot0, const int axor) {
    h_epi_quad<I, J, 0>(bias, hp, is_res, wrow, slot0, axor);
    h_epi_quad<I, J, 1>(bias, hp, is_res, wrow, slot0, axor);
    h_epi_quad<I, J, 2>(bias, hp, is_res, wrow, slot0, axor);
    h_epi_quad<I, J, 3>(bias, hp, is_res, wrow, slot0, axor);
}

__global__ __launch_bounds__(H_THREADS) void mlp_fwd128_kernel(FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    // the accumulator file is this kernel's: a[0:255] are written by name in the MFMA statements
    asm volatile("" ::: "a0", "a15", "a16", "a31", "a32", "a63", "a64", "a95", "a96", "a127", "a128", "a159", "a160", "a191", "a192", "a223",
                 "a224", "a255");
    char* const Abuf = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wvu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = blockIdx.x * H_BM;
    const unsigned mask = __builtin_amdgcn_readfirstlane((unsigned)p.tile_mask[m0 / SCENERF_TILE_ROWS] & 31u);
    const int* const Dg = p.desc + mask * F_MAXCH;
    const int nch = __builtin_amdgcn_readfirstlane(Dg[0]);   // header: number of chunks (a multiple of H_D per layer); descriptors follow

    // ---- one-time LDS contents: the biases of all seven layers, this mask's descriptors
    for (int i = tid; i < 7 * 128; i += H_THREADS) *(float4*)(lds + H_BIAS + i * 16) = *(const float4*)(p.layer[i >> 7].bias + (i & 127) * 4);
    int* const tab = (int*)(lds + H_TAB);
    for (int i = tid; i < nch + 16; i += H_THREADS) tab[i] = Dg[1 + i];

    // ---- per-lane values.  Everything derived from the lane index is RE-derived from an opaque copy at the top of every group of
    // chunks and of every epilogue: the 256 architectural registers are spoken for (residual stream 128, weight ring 64, fragments
    // 16), and a compiler that hoists a few dozen loop-invariant addresses parks them in accumulator registers -- which here hold
    // the accumulators (a[0:255] are written by name; tools/asmcheck.sh refuses a build with a compiler-made v_accvgpr_*).
    const char* const Wb = (const char*)p.Wst;
    const char* const gX3 = (const char*)p.X3;
    const char* const gZ = (const char*)p.Z;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    int ln = lane;   // the opaque copy (refreshed by H_LANE)
#define H_LANE() asm volatile("" : "+v"(ln))
    // (per group of chunks: byte offset of this lane's 16 bytes inside a w_stream block -- tile j adds 1 KiB -- and inside an A-buffer row)
    unsigned wl = 0, al = 0;
#define H_LANE_CONSTS()                                                                                          \
    H_LANE();                                                                                                    \
    wl = (unsigned)(((wvu * 128 + (ln & 31)) * 2 + ((ln >> 5) ^ ((ln >> 3) & 1))) * 16);                         \
    al = lds0 + (unsigned)((ln & 31) * F_AROW);
    // Activation fragments of the chunk described by d (wave-uniform): always 16 bytes from LDS at a_lds(d) + i * a_stride(d) for row tile
    // i -- the resident operand (slot (column / 8) + hi, XOR row & 15; 32 KiB between row tiles) or a streamed chunk's stage (row * 32 B,
    // 16-byte halves swapped when (row >> 3) & 1; 1 KiB between row tiles).
    typedef const __attribute__((address_space(3))) hfrag* lds_frag_p;
    typedef const __attribute__((address_space(1))) hfrag* glb_frag_p;
    auto a_lds = [&](const int d) __attribute__((always_inline)) {
        const unsigned res = al + (unsigned)((((FD_Y(d) >> 3) + (ln >> 5)) ^ (ln & 15)) << 4);
        const unsigned stg = lds0 + H_STG + HD_STAGE(d) * 4096 + (unsigned)((ln & 31) * 32 + (((ln >> 5) ^ ((ln >> 3) & 1)) << 4));
        return FD_SRC(d) ? stg : res;
    };
    auto a_stride = [&](const int d) __attribute__((always_inline)) { return FD_SRC(d) ? 1024u : 32u * F_AROW; };
    auto ld_lds = [&](hfrag& f, const unsigned aoff, const unsigned stride, const int i) __attribute__((always_inline)) {
        f = *(lds_frag_p)(uintptr_t)(aoff + i * stride);
    };
    auto load_a_all = [&](hfrag (&f)[4], const int d) __attribute__((always_inline)) {
        const unsigned ao = a_lds(d), as = a_stride(d);
        ld_lds(f[0], ao, as, 0); ld_lds(f[1], ao, as, 1); ld_lds(f[2], ao, as, 2); ld_lds(f[3], ao, as, 3);
    };
    // DMA of a streamed chunk (X3 / Z columns FD_Y .. +15 of the block's 128 rows) into its stage: this wave's 32 rows = one 1-KiB piece;
    // lane -> row lane / 2, physical 16-byte slot lane & 1, fetching the logical slot physical ^ ((row >> 3) & 1) (swizzle on the source)
    auto dma = [&](const int d) __attribute__((always_inline)) {
        const int gm = min(m0 + 32 * wvu + (ln >> 1), p.M - 1);                // (rows past M: clamped, computed, dropped)
        const unsigned pls = (unsigned)(((ln & 1) ^ ((ln >> 4) & 1)) << 4);
        const bool x3 = FD_SRC(d) == 1;
        const unsigned voff = (unsigned)gm * (x3 ? 3u * SCENERF_D_XENC * 2u : SCENERF_D_LATENT * 2u) + pls;   // < 4 GiB up to 865k rows
        const char* sb = (x3 ? gX3 : gZ) + FD_Y(d) * 2;
        h_glds16(sb, voff, __builtin_amdgcn_readfirstlane(lds0 + H_STG + HD_STAGE(d) * 4096 + wvu * 1024));
    };
    // lane's 16 bytes of tile j of a 16-KiB w_stream block ([512 rows n][32 B], halves swapped when (n >> 3) & 1): + 1 KiB per tile
    auto load_w = [&](hfrag (&w)[4], const int d) __attribute__((always_inline)) {
        const char* b = Wb + (size_t)FD_Z(d) * 16384;   // (uniform base + 32-bit lane offset)
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *(glb_frag_p)(uintptr_t)(b + (wl + 1024u * j));
    };

    // ---- layer output -> HBM: the A buffer of the finished layer is streamed out one 16-byte piece per thread per chunk of the NEXT
    // layer (32 pieces: rows 4 s .. 4 s + 3, one full row per wave), rectified values + sign bits (fused.hip)
    int save_layer = -1;   // (wave-uniform state stays scalar: the layer whose output is being streamed out, and the next piece)
    int save_i = 32;
    auto save_piece = [&]() __attribute__((always_inline)) {
        const int sl = __builtin_amdgcn_readfirstlane(save_layer);
        const int row = 4 * save_i + wvu, slot = ln;
        char* const save_ptr = sl >= 0 ? (char*)p.layer[sl].save : nullptr;
        if (save_ptr && m0 + row < p.M) {
            uint8_t* const sign_ptr = p.layer[sl].sign;
            const uint4 v = *(const uint4*)(Abuf + row * F_AROW + ((slot ^ (row & 15)) << 4));
            h_store16(save_ptr + (size_t)(m0 + row) * (p.layer[sl].save_ld * 2) + slot * 16, v);
            if (sign_ptr) {
                uint32_t u = h_pk_min_u16(v.x, 0x00010001u);
                u |= h_pk_min_u16(v.y, 0x00010001u) << 2;
                u |= h_pk_min_u16(v.z, 0x00010001u) << 4;
                u |= h_pk_min_u16(v.w, 0x00010001u) << 6;
                h_store1(sign_ptr + (size_t)(m0 + row) * 64 + slot, (u | (u >> 15)) & 0xffu);
            }
        }
        ++save_i;
    };

    // ---- the residual stream of the wave's 128 x 128 tile: packed bf16, hp[i][j][k] = elements 2k, 2k+1 of tile (i, j)
    uint32_t hp[4][4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) hp[i][j][k] = 0u;

    auto epilogue = [&](const int layer) __attribute__((always_inline)) {
        const FusedLayer& L = p.layer[layer];
        while (save_i < 32) save_piece();
        // (MFMA results are visible to v_accvgpr_read only after the pipeline has drained: 16 passes)
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // every wave has finished reading the A buffer for this layer
        const bool is_res = L.kind != 1;  // residual layers: out = h + acc + bias, h = bf16(out) ; fc_0 layers: out = acc + bias
        H_LANE();
        const int hi = ln >> 5, axor = ln & 15;
        const char* bb = lds + H_BIAS + layer * 2048 + (wvu * 128 + 4 * hi) * 4;
        char* const wr0 = Abuf + (ln & 31) * F_AROW + 8 * hi;
#define H_EPI_J(J)                                                                                      \
    {                                                                                                   \
        float bias[16];                                                                                 \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                 \
            const float4 b = *(const float4*)(bb + (J * 32 + q * 8) * 4);                               \
            bias[4 * q] = b.x; bias[4 * q + 1] = b.y; bias[4 * q + 2] = b.z; bias[4 * q + 3] = b.w;     \
        }                                                                                               \
        h_epi_tile<0, J>(bias, hp[0][J], is_res, wr0, wvu * 16 + J * 4, axor);                          \
        h_epi_tile<1, J>(bias, hp[1][J], is_res, wr0 + 32 * F_AROW, wvu * 16 + J * 4, axor);            \
        h_epi_tile<2, J>(bias, hp[2][J], is_res, wr0 + 64 * F_AROW, wvu * 16 + J * 4, axor);            \
        h_epi_tile<3, J>(bias, hp[3][J], is_res, wr0 + 96 * F_AROW, wvu * 16 + J * 4, axor);            \
    }
        H_EPI_J(0) H_EPI_J(1) H_EPI_J(2) H_EPI_J(3)
#undef H_EPI_J
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // A buffer complete
        save_layer = layer;
        save_i = 0;
    };

    __syncthreads();   // biases, descriptors are in LDS
    // ---- prologue: weight ring (chunks 0 .. 3), the first two streamed chunks, fragments of chunk 0
    hfrag wr[H_D][4], af[4];
    int d0 = tab[0], d1 = tab[1], d2 = tab[2], d3 = tab[3];
    d0 = __builtin_amdgcn_readfirstlane(d0); d1 = __builtin_amdgcn_readfirstlane(d1);
    d2 = __builtin_amdgcn_readfirstlane(d2); d3 = __builtin_amdgcn_readfirstlane(d3);
    H_LANE_CONSTS()
    if (FD_SRC(d0)) dma(d0);
    if (FD_SRC(d1)) dma(d1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    load_w(wr[0], d0); load_w(wr[1], d1); load_w(wr[2], d2); load_w(wr[3], d3);
    load_a_all(af, d0);
    __builtin_amdgcn_sched_barrier(0);

    // One chunk (DC; DN1 / DN2 = the next two chunks' descriptors, DW = the chunk four ahead):
    //   * streamed-operand protocol (side effects only): chunk c+1's stage must have landed and be visible before this chunk's rows
    //     prefetch its fragments -- wait for this wave's DMA (issued one chunk ago; younger than it: the four weight loads of that
    //     chunk, possibly two stores), barrier (all four pieces; also: everyone is done reading the stage chunk c+2 goes to) -- then
    //     chunk c+2's DMA is issued;
    //   * row tile i's four MFMAs, then -- its fragment register being free once they are issued -- the same row tile's fragment of
    //     the NEXT chunk: a single fragment set, each piece re-loaded 12 MFMAs (384 cycles) before its next use.  At a layer's last
    //     chunk that prefetch reads the old A buffer; the fragments are loaded again after the epilogue;
    //   * the ring slot's next weights, one piece of the previous layer's output on its way to HBM.
    // Nothing in here merges two definitions of a loaded register: the only branches are around MFMAs and around stores.
#define H_ROW(I, S, ZERO, SKIP)                                                     \
    if (!(SKIP)) {                                                                  \
        if (ZERO) h_row<true, I>(wr[S], af[I]);                                     \
        else h_row<false, I>(wr[S], af[I]);                                         \
    }                                                                               \
    if (I == 0) { ao_ = a_lds(dn1_); as_ = a_stride(dn1_); }                        \
    ld_lds(af[I], ao_, as_, I);                                                     \
    __builtin_amdgcn_sched_barrier(0);
#define H_STEP(S, DC, DN1, DN2, DW)                                                 \
    {                                                                               \
        const int dn1_ = DN1;                                                       \
        if (FD_SRC(dn1_)) {                                                         \
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                        \
            __builtin_amdgcn_s_barrier();                                           \
        }                                                                           \
        if (FD_SRC(DN2)) dma(DN2);                                                  \
        const bool zero_ = (S == 0) && FD_BEGIN(DC), skip_ = HD_SKIP(DC) != 0;      \
        unsigned ao_, as_;                                                          \
        H_ROW(0, S, zero_, skip_) H_ROW(1, S, zero_, skip_) H_ROW(2, S, zero_, skip_) H_ROW(3, S, zero_, skip_) \
        load_w(wr[S], DW);                                                          \
        if (save_i < 32) save_piece();                                              \
        __builtin_amdgcn_sched_barrier(0);                                          \
    }
    int c = 0;
#pragma unroll 1
    while (c < nch) {
        // descriptors of the next group (chunks c + 4 .. c + 7; the table is zero-padded past the end)
        const int4 dn = *(const int4*)(tab + c + 4);
        const int n0 = __builtin_amdgcn_readfirstlane(dn.x), n1 = __builtin_amdgcn_readfirstlane(dn.y);
        const int n2 = __builtin_amdgcn_readfirstlane(dn.z), n3 = __builtin_amdgcn_readfirstlane(dn.w);
        H_LANE_CONSTS()
        H_STEP(0, d0, d1, d2, n0)
        H_STEP(1, d1, d2, d3, n1)
        H_STEP(2, d2, d3, n0, n2)
        H_STEP(3, d3, n0, n1, n3)   // (a layer ends only here)
        if (FD_END(d3)) {
            epilogue(FD_LAYER(d3));
            H_LANE_CONSTS()
            load_a_all(af, n0);     // the next layer's first fragments, from the A buffer just written
        }
        d0 = n0; d1 = n1; d2 = n2; d3 = n3;
        c += H_D;
        __builtin_amdgcn_sched_barrier(0);
    }
#undef H_STEP
#undef H_ROW
    H_LANE();
    while (save_i < 32) save_piece();
    if (p.logits) {
        // lin_out on the rectified H3 tile still resident in the A buffer (all waves are past the last epilogue's second barrier):
        // w_out (fp32, <= 8 KiB) is copied into the now idle stages, then 8 threads per row take 64 columns each, a butterfly adds the
        // partials; 32 rows per pass (same summation order as fused.hip)
        float* wl = (float*)(lds + H_STG);
        for (int i = wvu * 64 + ln; i < p.d_out * (SCENERF_D_HIDDEN / 4); i += H_THREADS) *(float4*)(wl + i * 4) = *(const float4*)(p.w_out + i * 4);
        __syncthreads();
#pragma unroll 1
        for (int rb = 0; rb < 4; ++rb) {
            const int t = wvu * 64 + ln;
            const int row = rb * 32 + (t >> 3), part = t & 7;
            float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
            for (int s8 = 0; s8 < 8; ++s8) {
                const int slot = part * 8 + s8;
                const uint4 v = *(const uint4*)(Abuf + row * F_AROW + ((slot ^ (row & 15)) << 4));
                const float f[8] = {bf16lo(v.x), bf16hi(v.x), bf16lo(v.y), bf16hi(v.y), bf16lo(v.z), bf16hi(v.z), bf16lo(v.w), bf16hi(v.w)};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < p.d_out) {
                        const float4 w0 = *(const float4*)(wl + j * SCENERF_D_HIDDEN + slot * 8);
                        const float4 w1 = *(const float4*)(wl + j * SCENERF_D_HIDDEN + slot * 8 + 4);
                        o[j] = fmaf(f[0], w0.x, o[j]); o[j] = fmaf(f[1], w0.y, o[j]); o[j] = fmaf(f[2], w0.z, o[j]); o[j] = fmaf(f[3], w0.w, o[j]);
                        o[j] = fmaf(f[4], w1.x, o[j]); o[j] = fmaf(f[5], w1.y, o[j]); o[j] = fmaf(f[6], w1.z, o[j]); o[j] = fmaf(f[7], w1.w, o[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[j] += __shfl_xor(o[j], 1);
                o[j] += __shfl_xor(o[j], 2);
                o[j] += __shfl_xor(o[j], 4);
            }
            if (part == 0 && m0 + row < p.M) {
                for (int j = 0; j < p.d_out; ++j) p.logits[(size_t)(m0 + row) * p.d_out + j] = o[j] + p.b_out[j];
            }
        }
    }
}

// ---- host: chunk descriptors for the 32 scale masks, every layer padded to a multiple of H_D chunks with no-op chunks
int fwd128_table_build(const scenerf_cfg* cfg, std::vector<int>& tab) {
    tab.assign((size_t)32 * F_MAXCH, 0);
    int seg_off[5], off = 0;
    for (int i = 0; i < 5; ++i) { seg_off[i] = off; off += cfg->map_C[i]; }
    SRF_CHECK(off == SCENERF_D_LATENT, "fwd128: map channels do not add up to the latent width");
    const int layer_k[7] = {3 * SCENERF_D_XENC + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN,
                            SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN};
    int layer_block0[7], nb = 0;
    for (int i = 0; i < 7; ++i) { layer_block0[i] = nb; nb += layer_k[i] / F_BK; }
    SRF_CHECK(nb < 1024, "fwd128: w_stream block index does not fit the descriptor");
    for (int mask = 0; mask < 32; ++mask) {
        int* ch = tab.data() + (size_t)mask * F_MAXCH + 1;   // entry 0 is the header
        int n = 0;
        bool ok = true;
        auto seg = [&](int layer, int src, int a0, int w0, int len) {
            if (len % F_BK || a0 % F_BK || w0 % F_BK) ok = false;
            for (int k = 0; k + F_BK <= len; k += F_BK) {
                if (n >= F_MAXCH - 24) { ok = false; return; }
                ch[n] = (layer_block0[layer] + (w0 + k) / F_BK) | (((a0 + k) / F_BK) << 10) | (src << 18) | (layer << 20) | ((n % H_NSTG) << 26);
                ++n;
            }
        };
        auto zsegs = [&](int layer, int wbase) {
            for (int i = 0; i < 5; ++i) {
                if ((mask >> i) & 1) seg(layer, 2, seg_off[i], wbase, cfg->map_C[i]);
                wbase += cfg->map_C[i];
            }
        };
        auto pad = [&](int layer) {   // no-op chunks up to a multiple of H_D: resident operand, block 0, MFMAs skipped
            while (n % H_D) ch[n++] = (layer << 20) | (1 << 25);
        };
        seg(0, 1, 0, 0, 3 * SCENERF_D_XENC);
        zsegs(0, 3 * SCENERF_D_XENC);
        pad(0);
        for (int b = 0; b < 3; ++b) {
            seg(1 + 2 * b, 0, 0, 0, SCENERF_D_HIDDEN);
            pad(1 + 2 * b);
            seg(2 + 2 * b, 0, 0, 0, SCENERF_D_HIDDEN);
            if (b < 2) zsegs(2 + 2 * b, SCENERF_D_HIDDEN);
            pad(2 + 2 * b);
        }
        SRF_CHECK(ok && n % H_D == 0, "fwd128: segment lengths must be multiples of 16 and fit the descriptor table");
        for (int i = 0; i < n; ++i) {
            if (i + 1 == n || FD_LAYER(ch[i + 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 23;
            if (i == 0 || FD_LAYER(ch[i - 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 24;
        }
        ch[-1] = n;   // entries n .. stay zero: prefetches past the end read block 0 / the resident operand and are never used
    }
    return 0;
}

static SrfDescCache g_fwd128_table;

static int fwd128_attrs() {
    SRF_ONCE_PER_DEVICE(SRF_HIP(hipFuncSetAttribute((const void*)mlp_fwd128_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS)));
    return 0;
}

int fwd128_prepare(const scenerf_cfg* cfg, hipStream_t s) {
    if (int e = fwd128_attrs()) return e;
    const int* d = nullptr;
    return srf_desc_cache_get(g_fwd128_table, cfg, s, fwd128_table_build, &d);
}

int launch_mlp_fwd_128(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                       const scenerf_mlp_acts* a, hipStream_t s) {
    if (int e = fwd128_attrs()) return e;
    FusedArgs p = {};
    const int H = SCENERF_D_HIDDEN;
    const size_t sign_layer = (size_t)cdiv(M, SCENERF_TILE_ROWS) * SCENERF_TILE_ROWS * 64;
    auto sign = [&](int l) { return a->sign_bits ? a->sign_bits + l * sign_layer : nullptr; };
    p.layer[0] = {w->b_h[0], a->H[0], sign(0), 0, H};
    for (int b = 0; b < 3; ++b) {
        p.layer[1 + 2 * b] = {w->b_fc0[b], a->Nn[b], sign(1 + 2 * b), 1, H};
        p.layer[2 + 2 * b] = {w->b_h[b + 1], a->H[b + 1], b < 2 ? sign(2 + 2 * b) : nullptr, 2, H};
    }
    p.Wst = w->w_stream;
    p.X3 = a->h0pre;
    p.Z = Z;
    p.tile_mask = tile_mask;
    if (int e = srf_desc_cache_get(g_fwd128_table, cfg, s, fwd128_table_build, &p.desc)) return e;
    p.M = M;
    p.w_out = w->w_out;
    p.b_out = w->b_out;
    p.logits = a->logits;
    p.d_out = w->d_out;
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
    mlp_fwd128_kernel<<<cdiv(M, H_BM), H_THREADS, H_LDS, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_fwd128_kernel");
    return 0;
}
