// EXPERIMENT, NOT BUILT, NEVER RUN -- stream128_wip.hip (validated, see its header) plus the "fast run" of csrc/stream.hip carried over to
// 128-row blocks: the first 16 chunks of every resident-operand layer as a tight loop (4 LDS reads + 2 loads + one v_readlane per
// 8 MFMAs).  Compiles for gfx950 at 256 VGPRs with 14 spilled registers (60 B of scratch): the register budget is the open issue
// (128 accumulators + 8-register address table + ring + fragments).  This is the combination DESIGN.md §5 argues for: half the
// weight bytes per FLOP AND few non-MFMA instructions per MFMA.
// Register-streamed fused ResnetFC forward for gfx950 (bf16 operands), 128-row blocks: the whole 7-GEMM trunk (lin_in + lin_z.0, three
// residual blocks fc_0 / fc_1 + lin_z.b) and lin_out in ONE kernel.  reference scenerf/models/resnetfc.py:133-164.
//
// Same arithmetic, data layouts and saved activations as the LDS-ring kernel in fused.hip (results are bit-identical).  What bounds a
// 64-row block there is the weight stream: every workgroup needs all 16 KiB of a K chunk's weights for 256 MFMA cycles and a CU pulls
// only ~32 B/clk through its vector memory path -- so this kernel halves the weight bytes per FLOP:
//   * one workgroup = 128 rows (= one scale-mask tile), 8 waves; wave w owns output columns [64 w, 64 w + 64) for all 128 rows:
//     4 x 2 MFMA 32x32x16 tiles = 128 accumulators, 8 MFMAs per 16-wide K chunk;
//   * NO weight ring in LDS: the wave streams exactly its slice of w_stream (two coalesced 1-KiB global_load_dwordx4 per chunk)
//     straight into a 4-deep VGPR ring with plain loads -- hipcc counts vmcnt for them by itself once a scheduling barrier per step
//     stops it from sinking them to their first use -- so a hidden layer's K loop runs 32 chunks without a barrier.  No other load
//     kind may sit in that loop (an LDS-DMA makes hipcc wait vmcnt(0)): the streamed operand of the lin_in / lin_z segments (X3 / Z
//     rows, 4 KiB per chunk, shared by all waves) is register-staged by the two wave quads into four LDS stages, only streamed
//     chunks start with a barrier, and the activation / sign-bit / residual stores are inline asm (a store the compiler knows about
//     would turn every counted wait into vmcnt(0); hidden stores only make a counted wait more conservative);
//   * the resident A operand (relu of the previous layer, 128 KiB) fills LDS, and 128 accumulators + ring + fragments fill the
//     256 registers a wave has at 2 waves per SIMD, so the residual stream h cannot stay on the chip: it is parked in HBM as bf16 --
//     rounded exactly where the other kernels round it -- in the wave's own register layout (`hres`: 16 coalesced 16-byte accesses
//     per lane, written by the layers that produce h and read back by the next one that adds to it: +0.95 GB of traffic per pass at
//     M = 153,600, L2-warm);
//   * layer ends fall on multiples of four chunks (host-padded with no-op chunks) so that the ring slot is static in the 4x unrolled
//     loop and the epilogue has ONE site; the descriptors sit in LDS (64 at a time into one VGPR + v_readlane).
#include "fused.h"
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_f;

#define S_BM 128
#define S_THREADS 512
#define S_D 4                                   // weight ring depth in chunks (= chunks per group of the unrolled loop)
#define S_NW 12                                 // descriptor window: chunks c .. c + 11 (a group of four joins inside one 64-entry block)
#define S_ABUF (S_BM * F_AROW)                  // 131072
#define S_NSTG 3                                // streamed-operand stages
#define S_STGB (S_BM * F_BK * 2)                // 4096: 128 rows x 32 B
#define S_STG S_ABUF
#define S_BIAS (S_STG + S_NSTG * S_STGB)        // 143360: 7 layers x 2 KiB
#define S_TAB (S_BIAS + 7 * 2048)               // 157696: this tile mask's chunk descriptors (+ read slack)
#define S_LDS (S_TAB + 768 * 4)                 // 160768 of 163840
// descriptor bits as in fused.h except [25] = no-op chunk (padding: loads happen, MFMAs do not) and [26:27] = stage (chunk mod 4)
#define SD_SKIP(d) (((d) >> 25) & 1)
#define SD_STAGE(d) (((d) >> 26) & 3)
#define SD_FAST(d) (((d) >> 30) & 1)            // first chunk of a layer whose first S_FASTN chunks may run as one tight loop
#define S_FASTN 16

__device__ static inline void s_store16(void* p, uint4 v) {
    const u32x4_f t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ static inline void s_store1(void* p, uint32_t v) { asm volatile("global_store_byte %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

typedef unsigned short s_ushort2 __attribute__((ext_vector_type(2)));
__device__ static inline uint32_t s_pk_min_u16(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(s_ushort2, a), __builtin_bit_cast(s_ushort2, b)));
}

__global__ __launch_bounds__(S_THREADS) void mlp_stream_kernel(FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const Abuf = lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int m0 = blockIdx.x * S_BM;
    const unsigned mask = __builtin_amdgcn_readfirstlane((unsigned)p.tile_mask[m0 / SCENERF_TILE_ROWS] & 31u);
    desc_ptr D = (desc_ptr)(uintptr_t)(p.desc + mask * F_MAXCH);
    const int nch = D[0];   // header: number of chunks (a multiple of 4); the descriptors follow, zero-padded
    ++D;
    int* const tab = (int*)(lds + S_TAB);
    for (int i = tid; i < nch + S_NW + 8; i += S_THREADS) tab[i] = D[i];
    for (int i = tid; i < 7 * 128; i += S_THREADS)
        *(float4*)(lds + S_BIAS + i * 16) = *(const float4*)(p.layer[i >> 7].bias + (i & 127) * 4);

    // ---- weights: lane's 16 bytes of tile j of a 16-KiB w_stream block ([512 rows n][32 B], halves swapped when (n >> 3) & 1)
    const uint4* const Wb = (const uint4*)p.Wst;
    const int woff = (wvu * 64 + (lane & 31)) * 2 + ((lane >> 5) ^ ((lane >> 3) & 1));   // + 64 for tile 1
    // ---- streamed operand (X3 / Z rows): a quad's four waves cover rows 32 (w & 3) .. + 31; lane -> row lane / 2, physical slot lane & 1
    const int quad = wvu >> 2;
    const int gm_a = min(m0 + 32 * (wvu & 3) + (lane >> 1), p.M - 1);
    const int pls = ((lane & 1) ^ ((lane >> 4) & 1)) << 4;
    const unsigned ox3 = (unsigned)gm_a * (3 * SCENERF_D_XENC * 2) + pls;   // < 4 GiB: M * 4960 B fits 32 bits up to 865k rows
    const unsigned oz = (unsigned)gm_a * (SCENERF_D_LATENT * 2) + pls;
    auto s_load = [&](const int d) __attribute__((always_inline)) -> uint4 {   // (wave-uniform d with src != 0)
        const char* base = FD_SRC(d) == 1 ? (const char*)p.X3 : (const char*)p.Z;
        return *(const uint4*)(base + ((FD_SRC(d) == 1 ? ox3 : oz) + (unsigned)FD_Y(d) * 2));
    };
    auto s_write = [&](const int d, const uint4 v) __attribute__((always_inline)) {
        *(uint4*)(lds + S_STG + SD_STAGE(d) * S_STGB + (wvu & 3) * 1024 + lane * 16) = v;
    };

    // ---- fragments.  Transposed accumulator tile (i, j): lane holds activation row m = 32 i + (lane & 31) and outputs
    // n = 64 w + 32 j + 8 q + 4 (lane >> 5) + e in register 4 q + e.
    f32x16_f acc[4][2];
    const int arow = (lane & 31) * F_AROW;
    const int axor = lane & 15;
    const int offA2 = (lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4);
    auto init_acc = [&](const int layer) __attribute__((always_inline)) {   // accumulators start from the layer's bias (LDS copy)
        const char* bb = lds + S_BIAS + layer * 2048 + (wvu * 64 + 4 * (lane >> 5)) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *(const float4*)(bb + (j * 32 + q * 8) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i][j][4 * q] = b.x; acc[i][j][4 * q + 1] = b.y; acc[i][j][4 * q + 2] = b.z; acc[i][j][4 * q + 3] = b.w; }
            }
    };
    // the residual stream of this lane in HBM: [workgroup][wave][lane][4 i][2 j][4 q] x 4 bf16 (8 B) = 256 B per lane
    char* const hres = (char*)p.dH3 + (((size_t)blockIdx.x * 8 + wvu) * 64 + lane) * 256;

    // ---- layer output -> HBM: the A buffer of the finished layer is streamed out one 16-byte piece per thread per step of the
    // NEXT layer (16 pieces), stores and sign bytes by inline asm
    char* save_ptr = nullptr;
    uint8_t* sign_ptr = nullptr;
    int save_ld2 = 0;
    int save_i = 16;
    auto save_piece = [&]() __attribute__((always_inline)) {   // rows 8 i .. 8 i + 7: thread t moves slot t & 63 of row 8 i + t / 64
        const int row = 8 * save_i + (tid >> 6), slot = tid & 63;
        if (save_ptr && m0 + row < p.M) {
            const uint4 v = *(const uint4*)(Abuf + row * F_AROW + ((slot ^ (row & 15)) << 4));
            s_store16(save_ptr + (size_t)(m0 + row) * save_ld2 + slot * 16, v);
            if (sign_ptr) {   // 8 sign bits per piece (rectified values: positive == non-zero), see fused.hip
                uint32_t u = s_pk_min_u16(v.x, 0x00010001u);
                u |= s_pk_min_u16(v.y, 0x00010001u) << 2;
                u |= s_pk_min_u16(v.z, 0x00010001u) << 4;
                u |= s_pk_min_u16(v.w, 0x00010001u) << 6;
                s_store1(sign_ptr + (size_t)(m0 + row) * 64 + slot, (u | (u >> 15)) & 0xffu);
            }
        }
        ++save_i;
    };

    // ---- layer epilogue: out = [h +] acc (bias included) -> bf16 ; h parked in HBM ; relu(out) -> resident A buffer
    auto epilogue = [&](const int layer) __attribute__((always_inline)) {
        const FusedLayer& L = p.layer[layer];
        while (save_i < 16) save_piece();
        const bool is_res = L.kind != 1;      // residual layers (the first one too: h starts at 0) ; fc_0 layers: out = acc
        const bool has_h = is_res && layer > 0;
        // residual values of the lane, two row tiles at a time: tiles 0, 1 are fetched before the barrier (their latency hides behind
        // the wait for the slowest wave), tiles 2, 3 while tile 0 is converted
        uint4 hall[4][4];
        auto fetch = [&](const int i) __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 4; ++t) hall[i][t] = has_h ? *(const uint4*)(hres + i * 64 + t * 16) : uint4{0, 0, 0, 0};
        };
        if (has_h) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this lane's own residual stores of two layers ago (long done)
        fetch(0);
        fetch(1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // every wave has finished reading the A buffer for this layer
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 1) { fetch(2); fetch(3); }
            uint4 (&hv)[4] = hall[i];
            int wbase = (32 * i + (lane & 31)) * F_AROW + 8 * (lane >> 5);
            asm volatile("" : "+v"(wbase));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    const uint4 h4 = hv[j * 2 + qq];
                    const uint32_t hw[4] = {h4.x, h4.y, h4.z, h4.w};
                    uint32_t pk[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {   // q = 2 qq + (u >> 1), pair e = u & 1: elements 2 k, 2 k + 1 with k = 2 q + e
                        const int k = 4 * qq + u;
                        const float v0 = bf16lo(hw[u]) + acc[i][j][2 * k];      // (h is 0 where there is none: exact)
                        const float v1 = bf16hi(hw[u]) + acc[i][j][2 * k + 1];
                        pk[u] = pack_bf16x2(v0, v1);
                    }
                    if (is_res && layer < 6) s_store16(hres + i * 64 + (j * 2 + qq) * 16, uint4{pk[0], pk[1], pk[2], pk[3]});
#pragma unroll
                    for (int h2 = 0; h2 < 2; ++h2) {
                        const int slot = wvu * 8 + j * 4 + 2 * qq + h2;
                        uint2 o;
                        o.x = relu_bf16x2(pk[2 * h2]);
                        o.y = relu_bf16x2(pk[2 * h2 + 1]);
                        *(uint2*)(Abuf + wbase + ((slot ^ axor) << 4)) = o;
                    }
                }
        }
        asm volatile("" ::: "memory");
        init_acc(layer < 6 ? layer + 1 : 6);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();         // A buffer complete
        save_ptr = (char*)L.save;
        sign_ptr = L.sign;
        save_ld2 = L.save_ld * 2;
        save_i = 0;
    };

    // ---- prologue: descriptor window, streamed chunks 0..4, weight ring (chunks 0..3)
    // (named scalars, not an array: hipcc turns a select chain over array elements into a dynamically indexed load and the array
    //  moves to scratch memory)
    int q0 = D[0], q1 = D[1], q2 = D[2], q3 = D[3], q4 = D[4], q5 = D[5], q6 = D[6], q7 = D[7], q8 = D[8], q9 = D[9], q10 = D[10], q11 = D[11];
    // staging: quad u in {0, 1}.  In step 0 of a group it writes chunk c + 1 + u and fetches chunk c + 5 + u (register zrA), in step 2
    // likewise (register zrB) -- every chunk is staged once, >= 1 step before its use and 4 steps after its fetch; one register per
    // site, so hipcc sees the ring loads of four steps between a fetch and its write and keeps the wait counted
    uint4 zrA = {0, 0, 0, 0}, zrB = {0, 0, 0, 0};
    {   // chunk 0: straight into its stage (quad 0); chunks 1, 2 -> zrA of quads 0, 1 ; chunks 3, 4 -> zrB
        if (quad == 0 && FD_SRC(q0)) s_write(q0, s_load(q0));
        const int da = quad == 0 ? q1 : q2, db = quad == 0 ? q3 : q4;
        if (FD_SRC(da)) zrA = s_load(da);
        if (FD_SRC(db)) zrB = s_load(db);
    }
    // (the ring loads come AFTER the staging loads: the first staging write then has a full ring of younger loads in front of it on
    // every path, like in the steady state, and hipcc's counted wait there does not drain the ring)
    uint4 ring[S_D][2];
#pragma unroll
    for (int s = 0; s < S_D; ++s) {
        const uint4* b = Wb + (size_t)FD_Z(s == 0 ? q0 : s == 1 ? q1 : s == 2 ? q2 : q3) * 1024 + woff;
        ring[s][0] = b[0];
        ring[s][1] = b[64];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // descriptors, stage 0
    init_acc(0);
    int dv = tab[lane];                   // descriptors 64 k .. 64 k + 63 of the block the window's head (c + S_NW) is in
    __builtin_amdgcn_sched_barrier(0);

    int c = 0;
    int dend = 0;                         // descriptor of the last chunk of the group just done
    auto step = [&](auto SC) __attribute__((always_inline)) {
        constexpr int S = decltype(SC)::value;
        const int idx = c + S_NW;         // joins the window at the end of the step
        if ((idx & 63) == 0) dv = tab[idx + lane];
        const int dn = __builtin_amdgcn_readlane(dv, idx & 63);
        const int d0 = q0;
        if (FD_SRC(d0)) {                 // streamed chunk: its stage was written at least one step ago by a quad
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        if (!SD_SKIP(d0)) {
            uint4 a[4];
            if (FD_SRC(d0) == 0) {
                const int kslot = (FD_Y(d0) >> 3) + (lane >> 5);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(Abuf + i * 32 * F_AROW + arow + ((kslot ^ axor) << 4));
            } else {
                const char* St = lds + S_STG + SD_STAGE(d0) * S_STGB + offA2;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(St + i * 1024);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)   // C^T tile: rows = outputs n, cols = activation rows m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_f, ring[S][j]), __builtin_bit_cast(bf16x8_f, a[i]),
                                                                        acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        {   // refill this ring slot with chunk c + 4
            const uint4* b = Wb + (size_t)FD_Z(q4) * 1024 + woff;
            ring[S][0] = b[0];
            ring[S][1] = b[64];
        }
        if (S == 0) {                     // staging site A: quad u writes chunk c + 1 + u, fetches chunk c + 5 + u
            const int dw = quad == 0 ? q1 : q2, dl = quad == 0 ? q5 : q6;
            if (FD_SRC(dw)) s_write(dw, zrA);
            if (FD_SRC(dl)) zrA = s_load(dl);
        }
        if (S == 2) {                     // staging site B, two steps later
            const int dw = quad == 0 ? q1 : q2, dl = quad == 0 ? q5 : q6;
            if (FD_SRC(dw)) s_write(dw, zrB);
            if (FD_SRC(dl)) zrB = s_load(dl);
        }
        if (save_i < 16) save_piece();
        if (S == S_D - 1) dend = d0;      // (layer ends only here: the epilogue runs after the group, at its single site)
        q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5; q5 = q6; q6 = q7; q7 = q8; q8 = q9; q9 = q10; q10 = q11; q11 = dn;
        ++c;
        __builtin_amdgcn_sched_barrier(0);   // nothing moves across a step: the ring loads stay where they are written
    };
    // fast run (see csrc/stream.hip): the first S_FASTN chunks of a resident-operand layer as a tight loop
    int aoff[8];                          // [chunk & 7]: row base + swizzled 16-byte slot of this lane
#pragma unroll
    for (int k = 0; k < 8; ++k) aoff[k] = arow + (((2 * k + (lane >> 5)) ^ axor) << 4);
    auto fast_run = [&]() __attribute__((always_inline)) {
        const int dvf = tab[c + lane];
#pragma unroll
        for (int t = 0; t < S_FASTN; ++t) {
            uint4 a[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = *(const uint4*)(Abuf + i * 32 * F_AROW + (t >> 3) * 256 + aoff[t & 7]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_f, ring[t & 3][j]), __builtin_bit_cast(bf16x8_f, a[i]),
                                                                        acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            const uint4* b = Wb + (size_t)FD_Z(__builtin_amdgcn_readlane(dvf, t + 4)) * 1024 + woff;
            ring[t & 3][0] = b[0];
            ring[t & 3][1] = b[64];
            save_piece();                 // pieces 0 .. 15 of the previous layer's output
            __builtin_amdgcn_sched_barrier(0);
        }
        c += S_FASTN;
        q0 = __builtin_amdgcn_readlane(dvf, S_FASTN + 0); q1 = __builtin_amdgcn_readlane(dvf, S_FASTN + 1);
        q2 = __builtin_amdgcn_readlane(dvf, S_FASTN + 2); q3 = __builtin_amdgcn_readlane(dvf, S_FASTN + 3);
        q4 = __builtin_amdgcn_readlane(dvf, S_FASTN + 4); q5 = __builtin_amdgcn_readlane(dvf, S_FASTN + 5);
        q6 = __builtin_amdgcn_readlane(dvf, S_FASTN + 6); q7 = __builtin_amdgcn_readlane(dvf, S_FASTN + 7);
        q8 = __builtin_amdgcn_readlane(dvf, S_FASTN + 8); q9 = __builtin_amdgcn_readlane(dvf, S_FASTN + 9);
        q10 = __builtin_amdgcn_readlane(dvf, S_FASTN + 10); q11 = __builtin_amdgcn_readlane(dvf, S_FASTN + 11);
        dv = tab[((c + S_NW - 1) & ~63) + lane];
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    while (c < nch) {
        if (SD_FAST(q0)) fast_run();
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        if (FD_END(dend)) epilogue(FD_LAYER(dend));
        __builtin_amdgcn_sched_barrier(0);
    }
    while (save_i < 16) save_piece();
    if (p.logits) {
        // lin_out on the rectified H3 tile still resident in the A buffer (all waves are past the last epilogue's second barrier).
        // w_out (fp32, <= 8 KiB) goes into the idle stages first; then 4 threads per row take 128 columns each, a butterfly adds them
        float* wl = (float*)(lds + S_STG);
        if (tid < p.d_out * (SCENERF_D_HIDDEN / 4)) *(float4*)(wl + tid * 4) = *(const float4*)(p.w_out + tid * 4);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int row = tid >> 2, part = tid & 3;
        float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int s16 = 0; s16 < 16; ++s16) {
            const int slot = part * 16 + s16;
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
        }
        if (part == 0 && m0 + row < p.M) {
            for (int j = 0; j < p.d_out; ++j) p.logits[(size_t)(m0 + row) * p.d_out + j] = o[j] + p.b_out[j];
        }
    }
}

// chunk descriptors for the 32 possible scale masks (forward): like fused.hip's table, every layer padded to a multiple of four chunks
struct StreamTable {
    int seg_len[5] = {-1, -1, -1, -1, -1};
    int* d_desc = nullptr;
};
static StreamTable g_stream_table;

static int stream_table_get(const scenerf_cfg* cfg, hipStream_t s, const int** desc) {
    bool same = g_stream_table.d_desc != nullptr;
    for (int i = 0; i < 5; ++i) same = same && g_stream_table.seg_len[i] == cfg->map_C[i];
    if (!same) {
        std::vector<int> tab((size_t)32 * F_MAXCH, 0);
        int seg_off[5], off = 0;
        for (int i = 0; i < 5; ++i) { seg_off[i] = off; off += cfg->map_C[i]; }
        SRF_CHECK(off == SCENERF_D_LATENT, "stream mlp: map channels do not add up to the latent width");
        const int layer_k[7] = {3 * SCENERF_D_XENC + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN,
                                SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN};
        int layer_block0[7], nb = 0;
        for (int i = 0; i < 7; ++i) { layer_block0[i] = nb; nb += layer_k[i] / F_BK; }
        for (int mask = 0; mask < 32; ++mask) {
            int* ch = tab.data() + (size_t)mask * F_MAXCH + 1;
            int n = 0;
            bool ok = true;
            auto seg = [&](int layer, int src, int a0, int w0, int len) {
                if (len % F_BK || a0 % F_BK) ok = false;
                for (int k = 0; k + F_BK <= len; k += F_BK) {
                    if (n >= F_MAXCH - 24) { ok = false; return; }   // (all five scales: 680 chunks with the padding; the window reads 20 further)
                    ch[n] = (layer_block0[layer] + (w0 + k) / F_BK) | (((a0 + k) / F_BK) << 10) | (src << 18) | (layer << 20) | ((n % S_NSTG) << 26);
                    ++n;
                }
            };
            auto zsegs = [&](int layer, int wbase) {
                for (int i = 0; i < 5; ++i) {
                    if ((mask >> i) & 1) seg(layer, 2, seg_off[i], wbase, cfg->map_C[i]);
                    wbase += cfg->map_C[i];
                }
            };
            auto pad = [&](int layer) {   // no-op chunks up to a multiple of four: block 0, MFMAs skipped
                while (n % S_D) { ch[n] = (layer << 20) | (1 << 25) | ((n % S_NSTG) << 26); ++n; }
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
            SRF_CHECK(ok && n % S_D == 0, "stream mlp: segment lengths must be multiples of 16 and fit the descriptor table");
            for (int i = 0; i < n; ++i) {
                if (i + 1 == n || FD_LAYER(ch[i + 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 23;
                if (i == 0 || FD_LAYER(ch[i - 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 24;
            }
            // fast runs: the first S_FASTN chunks of a layer from the resident operand, no layer end inside, no staging work due
            // (sites at steps g and g + 2 write chunks up to g + 4 and fetch chunks up to g + 8)
            for (int f = 0; f + S_FASTN < n; f += 4) {
                if (!FD_BEGIN(ch[f])) continue;
                bool fast = true;
                for (int i = 0; i < S_FASTN; ++i) fast = fast && FD_SRC(ch[f + i]) == 0 && !SD_SKIP(ch[f + i]) && !FD_END(ch[f + i]);
                for (int i = 1; i <= S_FASTN + 6; ++i) fast = fast && FD_SRC(ch[f + i]) == 0;
                if (fast) ch[f] |= 1 << 30;
            }
            ch[-1] = n;   // the entries after n stay zero: prefetches past the end read block 0 and are never used
        }
        if (!g_stream_table.d_desc) SRF_HIP(hipMalloc((void**)&g_stream_table.d_desc, tab.size() * sizeof(int)));
        SRF_HIP(hipStreamSynchronize(s));
        SRF_HIP(hipMemcpy(g_stream_table.d_desc, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
        for (int i = 0; i < 5; ++i) g_stream_table.seg_len[i] = cfg->map_C[i];
    }
    *desc = g_stream_table.d_desc;
    return 0;
}

int launch_mlp_fwd_stream(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                          const scenerf_mlp_acts* a, hipStream_t s) {
    static bool attr_done = false;
    if (!attr_done) {
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS));
        attr_done = true;
    }
    SRF_CHECK(a->hres, "stream mlp: acts->hres (residual scratch, [ceil(M/128)*128][512] bf16) is NULL");
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
    p.dH3 = a->hres;   // (the forward has no use for the backward's field: it carries the residual scratch)
    p.tile_mask = tile_mask;
    if (int e = stream_table_get(cfg, s, &p.desc)) return e;
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
    mlp_stream_kernel<<<cdiv(M, S_BM), S_THREADS, S_LDS, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_stream_kernel");
    return 0;
}
