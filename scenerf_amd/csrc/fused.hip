// Fused ResnetFC kernels for gfx950 (bf16 operands).  MODE 0: ONE kernel evaluates the whole 7-GEMM forward trunk (lin_in + lin_z.0,
// then three residual blocks fc_0 / fc_1 + lin_z.b) and lin_out for a block of 64 rows; MODE 1: the 6-GEMM dgrad chain of the
// blocks in the backward pass, same pipeline (see the kernel's comment).  reference scenerf/models/resnetfc.py:41-57,133-164.
//
// Why: the per-layer GEMMs (gemm.hip) are HBM-limited -- a K = 512 hidden layer with bf16 activations in HBM has an
// arithmetic intensity of only ~170 FLOP/B.  Here the 512-wide residual stream never leaves the chip:
//   * 8 consumer waves (2 along M x 4 along N, wave tile 32 x 128) hold the fp32 accumulators and the residual stream h
//     (packed bf16, rounded exactly where the layer path rounds H_b) in VGPRs;
//   * the A operand of the next GEMM, relu(x) in bf16, is written by the layer epilogue straight into a 64 KiB LDS-resident
//     buffer (XOR-swizzled 16-byte slots, conflict-free fragment reads) -- the very bytes the backward pass wants saved
//     (relu(H_b), relu(N_b): masks and wgrad operands only ever use the rectified value), so each activation is written
//     to HBM once, from LDS, with coalesced 16-byte stores, and never read back in the forward;
//   * 4 producer waves (one per SIMD) do nothing but stream: the weights come from L2 with global_load_lds through a
//     5-stage ring of 18 KiB (w_stream: pre-tiled so that a 1 KiB piece is contiguous memory and already the swizzled LDS
//     image; four pieces per M0 setup via the instruction offset), plus the gathered-feature / encoding chunks of the
//     lin_z / lin_in segments and the next layer's bias.  Only the producers count vmcnt; the consumers' stores never enter it.
// One raw s_barrier per 16-element K chunk; per-chunk bookkeeping is ONE 32-bit descriptor read with a scalar load (the first
// version spent ~45 SALU instructions per chunk per wave on cursors and was SALU-bound).  Scale segments a 128-row tile
// does not touch are skipped (tile_mask).  Measured history in DESIGN.md §5.
#include "fused.h"
#include <vector>
#include <cstdio>

#define F_WSTG (512 * F_BK * 2)           // W stage: 512 output columns x 32 B
#define F_STAGE (F_WSTG + F_A2STG)        // 18432
#define F_NST 5                           // ring depth
#define F_BIAS (F_ABUF + F_NST * F_STAGE) // 2 KiB: the next layer's bias
#define F_LDS (F_BIAS + 2048)             // 159744 of 163840
#define F_SIGN F_BIAS                     // backward: 64 rows x 64 B of sign bits (no bias there)
#define F_LDS_BWD (F_SIGN + 4096)          // 161792

// 16 bytes per lane, global -> LDS, no VGPR round trip: source = uniform base (SGPR pair) + 32-bit per-lane offset, destination
// = M0 (wave-uniform LDS address) + 16 * lane
__device__ static inline void f_glds16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}

// four consecutive 1 KiB pieces with ONE M0 setup: the instruction offset advances the global and the LDS address alike
__device__ static inline void f_glds16x4(const void* sbase, unsigned voff, unsigned lds_wave_base) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_wave_base)
                 : "memory");
}

__device__ static inline uint32_t pk_min_u16(uint32_t a, uint32_t b) {   // v_pk_min_u16, by name (see wide.hip: the compiler's own form of
    uint32_t r;                                                                   // min(x, 1) is compare + select per half)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}


#define F_THREADS 768   // 8 consumer waves (fragments + MFMA + epilogue) and 4 producer waves (one per SIMD: the weight stream)

// MODE 0: forward trunk (7 layers).  MODE 1: backward dgrad chain (6 layers: for b = 2, 1, 0: dN_b = (dH_{b+1} W1_b) * [N_b > 0],
// dH_b = dH_{b+1} + (dN_b W0_b) * [H_b > 0]; resnetfc.py:41-57 differentiated) -- same pipeline, the running gradient dH plays
// the residual stream's role, and the producers also turn the saved activations into sign bits for the consumers' epilogue.
template <int MODE>
__global__ __launch_bounds__(F_THREADS) void mlp_fused_kernel(FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* Abuf = lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int m0 = blockIdx.x * F_BM;
    const unsigned mask = MODE == 0 ? __builtin_amdgcn_readfirstlane((unsigned)p.tile_mask[m0 / SCENERF_TILE_ROWS] & 31u) : 0u;
    desc_ptr D = (desc_ptr)(uintptr_t)(p.desc + mask * F_MAXCH);
    const int nch = D[0];   // entry 0: header (total number of chunks); the descriptors follow
    ++D;
    const unsigned lds0 = (unsigned)(uintptr_t)lds;
    const unsigned ring0 = lds0 + F_ABUF;

    // Pipeline, one raw barrier per 16-wide K chunk (+ two per layer end).  In step c the consumers hold chunk c's fragments in
    // registers (read in step c-1), read chunk c+1's (landed) while their MFMAs of chunk c run; chunks c+2, c+3 are in flight
    // and the producers issue chunk c+4 into the stage chunk c-1 occupied (every consumer finished reading that one before its
    // MFMAs of c-1, i.e. before this barrier).  Ring: 5 stages.  Loads are counted by the waves that issue them: only the
    // producers wait on vmcnt, and the consumers' activation stores never enter those counts.
    if (wvu >= 8) {
        // ================================================================================ producers
        // glds pieces are 1 KiB.  W: contiguous in w_stream (already the LDS image); producer q fetches bytes [4096 q, +4096)
        // of the 16 KiB block, four pieces behind one M0 setup.  Streamed A (X3 / Z rows): piece = 32 rows x 32 B; lane -> (row
        // lane / 2, physical 16-byte slot lane & 1) fetching the logical slot physical ^ ((row >> 3) & 1) (swizzle on the SOURCE
        // address); producers 0 and 1.  Bias of a layer (2 KiB, with the layer's first chunk): producers 2 and 3.
        const int q = wvu - 8;
        const unsigned wlane = q * 4096 + lane * 16;
        const int prow = lane >> 1;
        const int pls = ((lane & 1) ^ ((lane >> 4) & 1)) << 4;
        // (row offsets RELATIVE to the block's first row -- the block's base goes into the 64-bit uniform address: as absolute 32-bit
        //  offsets they wrapped beyond 865,900 rows of Z (4,960 B each), i.e. in every no_grad chunk of more than 1,691 rays at N = 512 --
        //  found by tests/test_gpu_render.py::test_render_image_n512_at_the_benched_chunk_against_the_oracle, round 6)
        const int lr_a = min(m0 + 32 * (q & 1) + prow, p.M - 1) - m0;
        const unsigned ox3 = (unsigned)lr_a * (3 * SCENERF_D_XENC * 2) + pls;
        const unsigned oz = (unsigned)lr_a * (SCENERF_D_LATENT * 2) + pls;
        const char* const bx3 = (const char*)p.X3 + (size_t)m0 * (3 * SCENERF_D_XENC * 2);
        const char* const bz = (const char*)p.Z + (size_t)m0 * (SCENERF_D_LATENT * 2);
        auto issue = [&](const int d) {
            const unsigned sb = ring0 + FD_STAGE(d) * F_STAGE;
            f_glds16x4((const char*)p.Wst + (size_t)FD_Z(d) * F_WSTG, wlane, __builtin_amdgcn_readfirstlane(sb + q * 4096));
            if (MODE == 1) {
                // the sign bits gating this layer's output (written by the forward kernel): 16 rows x 64 B per producer, fetched with
                // the layer's 9th chunk -- issued while the layer's 5th chunk is computed, i.e. after the previous layer's epilogue
                // has read the old bits, and 23 steps before this layer's epilogue needs the new ones
                if (((d >> 10) & 255) == 8)
                    f_glds16(p.layer[FD_LAYER(d)].sign + (size_t)(m0 + 16 * q) * 64, lane * 16, __builtin_amdgcn_readfirstlane(lds0 + F_SIGN + q * 1024));
            }
            if (MODE == 0) {
                const int src = FD_SRC(d);
                if (src != 0 && q < 2)
                    f_glds16((src == 1 ? bx3 : bz) + (size_t)FD_Y(d) * 2, src == 1 ? ox3 : oz,
                             __builtin_amdgcn_readfirstlane(sb + F_WSTG + q * 1024));
                if (FD_BEGIN(d) && q >= 2)   // the bias this layer's accumulators start from (read at the previous layer's end)
                    f_glds16((const char*)p.layer[FD_LAYER(d)].bias + (q - 2) * 1024, lane * 16,
                             __builtin_amdgcn_readfirstlane(lds0 + F_BIAS + (q - 2) * 1024));
            }
        };
        if (MODE == 1) {
            // the incoming gradient tile dH3 -> resident A buffer: one row (1 KiB) per piece, 16 rows per producer; lane = physical
            // slot, fetching the logical slot lane ^ (row & 15) (the A buffer's swizzle)
#pragma unroll 1
            for (int i = 0; i < 16; ++i) {
                const int r = 16 * q + i;
                const int gr = min(m0 + r, p.M - 1);
                f_glds16((const char*)p.dH3 + (size_t)gr * p.dH_ld * 2, (unsigned)((lane ^ (r & 15)) << 4), __builtin_amdgcn_readfirstlane(lds0 + r * F_AROW));
            }
        }
#pragma unroll 1
        for (int c = 0; c < F_NST - 1 && c < nch; ++c) issue(D[c]);   // (forward: chunk 0 carries layer 0's bias)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int steps = (nch + 1) & ~1;   // the consumers run their ping-pong in pairs
        int d_iss = D[F_NST - 1], d_cur = D[0];   // chunk c + 4 (to issue), chunk c (layer ends: the epilogue has two more barriers)
        int c = 0;
        auto pstep = [&]() {
            const int d_iss_n = D[c + F_NST], d_cur_n = D[c + 1];   // next step's descriptors (the table is zero-padded)
            // chunk c+1 has landed once at most the loads of chunks c+2, c+3 (>= 4 per producer each) are outstanding; the sign row
            // fetched in step c-2 is older than those as well
            if (c + F_NST - 1 < nch) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tail
            __builtin_amdgcn_s_barrier();
            if (c + F_NST - 1 < nch) issue(d_iss);   // into the stage chunk c - 1 occupied
            if (FD_END(d_cur)) { __builtin_amdgcn_s_barrier(); __builtin_amdgcn_s_barrier(); }
            d_iss = d_iss_n;
            d_cur = d_cur_n;
            ++c;
        };
#pragma unroll 1
        while (c < steps) pstep();
        return;
    }

    // ==================================================================================== consumers
    const int wm = wv >> 2, wn = wv & 3;
    // fragment offsets inside a stage: W tile j adds j * 1024 (the swizzle term does not depend on j)
    const int ra = wm * 32 + (lane & 31);           // this lane's activation row inside the 64-row block
    const int offW = (wn * 128 + (lane & 31)) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4);
    const int offA2 = F_WSTG + ra * 32 + (((lane >> 5) ^ ((ra >> 3) & 1)) << 4);
    const int abase = ra * F_AROW;                  // resident A buffer: row base; 16-byte slot index is XORed with row & 15
    const int axor = ra & 15;

    // transposed accumulator tile j: lane holds activation row m = wm*32 + (lane & 31) and outputs n = wn*128 + 32 j + 8 q +
    // 4 (lane >> 5) + e in register r = 4 q + e.  Forward: the accumulators start from the layer's bias (LDS copy).  The residual
    // stream h (backward: the running gradient dH) is kept as packed bf16 pairs (elements 2i, 2i+1 of tile j in hp[j][i]): it is
    // rounded to bf16 at every block, exactly where the layer path rounds it when it writes H_b -- and it is what the A buffer
    // receives (forward: before the relu).
    f32x16_f acc[4];
    uint32_t hp[4][8];
    auto init_acc = [&]() {
        if (MODE == 0) {
            const char* bb = lds + F_BIAS + (wn * 128 + 4 * (lane >> 5)) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 b = *(const float4*)(bb + (j * 32 + q * 8) * 4);
                    acc[j][4 * q] = b.x; acc[j][4 * q + 1] = b.y; acc[j][4 * q + 2] = b.z; acc[j][4 * q + 3] = b.w;
                }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        }
    };

    struct Frags { uint4 a, b[4]; };
    // fragments of one chunk: W from its ring stage; the activation operand from the resident A buffer or the stage
    auto load_frags = [&](Frags& f, const int d) {
        const char* S = lds + F_ABUF + FD_STAGE(d) * F_STAGE;
        const int kslot = (FD_Y(d) >> 3) + (lane >> 5);
        const char* pa = (MODE == 1 || FD_SRC(d) == 0) ? Abuf + abase + ((kslot ^ axor) << 4) : S + offA2;
        f.a = *(const uint4*)pa;
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = *(const uint4*)(S + offW + j * 1024);
    };
    auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int j = 0; j < 4; ++j)   // C^T tile: rows = outputs n, cols = activation rows m
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_f, f.b[j]), __builtin_bit_cast(bf16x8_f, f.a), acc[j], 0, 0, 0);
    };
    // layer output -> HBM: the A buffer of the finished layer is streamed out one 16-byte piece per consumer thread per step of the
    // NEXT layer (8 in all)
    char* save_ptr = nullptr;
    uint8_t* sign_ptr = nullptr;
    int save_ld2 = 0;   // row stride in bytes
    int save_i = 8;
    auto save_piece = [&]() {   // rows 8 i .. 8 i + 7 of the block: thread t moves slot t & 63 of row 8 i + t / 64
        const int row = 8 * save_i + (tid >> 6), slot = tid & 63;
        if (save_ptr && m0 + row < p.M) {   // (inference: only the last layer has a destination)
            const uint4 v = *(const uint4*)(Abuf + row * F_AROW + ((slot ^ (row & 15)) << 4));
            *(uint4*)(save_ptr + (size_t)(m0 + row) * save_ld2 + slot * 16) = v;
            if (MODE == 0 && sign_ptr) {
                // the backward chain's gates: 8 sign bits per piece (the values are rectified: positive == non-zero; min(x, 1) per
                // half gives the bit), one byte store -- 64 consecutive bytes per row
                // min(x, 1) per half -> 0x000b000a per word; the four words interleave into 0x00BB00AA (b = odd, a = even elements)
                uint32_t u = pk_min_u16(v.x, 0x00010001u);
                u |= pk_min_u16(v.y, 0x00010001u) << 2;
                u |= pk_min_u16(v.z, 0x00010001u) << 4;
                u |= pk_min_u16(v.w, 0x00010001u) << 6;
                sign_ptr[(size_t)(m0 + row) * 64 + slot] = (uint8_t)(u | (u >> 15));
            }
        }
        ++save_i;
    };
    // ---- layer epilogue: residual in registers, output -> resident A buffer (-> HBM, see save_piece)
    auto epilogue = [&](int layer) {
        const FusedLayer& L = p.layer[layer];
        int wbase = ra * F_AROW + 8 * (lane >> 5);
        asm volatile("" : "+v"(wbase));   // keep the 16 swizzled addresses out of loop-invariant registers
        while (save_i < 8) save_piece();  // (only if a layer had fewer than 8 chunks)
        uint4 sg = {0, 0, 0, 0};
        if (MODE == 1) sg = *(const uint4*)(lds + F_SIGN + ra * 64 + wn * 16);   // sign bits of this lane's row, outputs wn*128 .. +127
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // every consumer has finished reading the A buffer for this layer
        // forward:  residual layers (the first one too: h starts at 0): out = h + acc, h = bf16(out) ; fc_0 layers: out = acc
        // backward: kind 1: out = acc * [N > 0] ; else: out = dh + acc * [H > 0], dh = bf16(out)
        // branch-free with a uniform 0/1 factor (exact)
        const bool is_res = L.kind != 1;
        const float resf = is_res ? 1.f : 0.f;
        const uint32_t sgw[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint32_t pk[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int i = 2 * q + e;   // elements 2i, 2i+1 ; output n - wn*128 = 32 j + 8 q + 4 hi + 2 e + {0, 1}
                    float a0 = acc[j][2 * i], a1 = acc[j][2 * i + 1];
                    if (MODE == 1) {           // gate by the sign bit: bit (8 q + 2 e) + 4 hi of word j ; v_bfe_i32 gives 0 / -1
                        const uint32_t w = sgw[j] >> (4 * (lane >> 5));
                        a0 = __uint_as_float(__float_as_uint(a0) & (uint32_t)__builtin_amdgcn_sbfe(w, 8 * q + 2 * e, 1));
                        a1 = __uint_as_float(__float_as_uint(a1) & (uint32_t)__builtin_amdgcn_sbfe(w, 8 * q + 2 * e + 1, 1));
                    }
                    const float v0 = __builtin_fmaf(bf16lo(hp[j][i]), resf, a0);      // forward: bias included in acc
                    const float v1 = __builtin_fmaf(bf16hi(hp[j][i]), resf, a1);
                    pk[e] = pack_bf16x2(v0, v1);
                    hp[j][i] = is_res ? pk[e] : hp[j][i];
                }
                const int slot = wn * 16 + j * 4 + q;
                uint2 o;   // forward: relu after rounding (the rounding keeps the sign): one v_pk_max_i16 per pair
                o.x = MODE == 0 ? relu_bf16x2(pk[0]) : pk[0];
                o.y = MODE == 0 ? relu_bf16x2(pk[1]) : pk[1];
                *(uint2*)(Abuf + wbase + ((slot ^ axor) << 4)) = o;   // four consecutive outputs: one 8-byte LDS write
            }
        asm volatile("" ::: "memory");   // bias reads go straight into the (now dead) accumulators, not into 64 temporaries
        init_acc();        // forward: next layer's bias (its DMA was issued with that layer's first chunk, which has landed)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // A buffer complete: the next layer may read it
        save_ptr = (char*)L.save;
        sign_ptr = L.sign;
        save_ld2 = L.save_ld * 2;
        save_i = 0;
    };

    Frags f0, f1;
    __builtin_amdgcn_s_barrier();    // producers: bias(0) / dH3 tile and chunks 0..3 have landed
    init_acc();
    if (MODE == 1) {                 // running gradient <- the dH3 tile, in accumulator layout (mirror of the epilogue's write)
        const int rbase = ra * F_AROW + 8 * (lane >> 5);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint2 v = *(const uint2*)(Abuf + rbase + (((wn * 16 + j * 4 + q) ^ axor) << 4));
                hp[j][2 * q] = v.x;
                hp[j][2 * q + 1] = v.y;
            }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) hp[j][i] = 0u;
    }
    int d_cur = D[0], d_nxt = D[1];  // chunk c (in registers), chunk c + 1 (fragments read during step c)
    load_frags(f0, d_cur);
    int c = 0;
    auto step = [&](Frags& cur, Frags& nxt) {
        const int d_nxt_n = D[c + 2];    // next step's descriptor (the table is zero-padded)
        __builtin_amdgcn_s_barrier();
        load_frags(nxt, d_nxt);          // (past the end: a harmless in-bounds read)
        mfmas(cur);
        if (save_i < 8) save_piece();
        if (FD_END(d_cur)) {
            epilogue(FD_LAYER(d_cur));
            // the next layer starts with the resident operand: its first activation fragment must see the A buffer just written
            const int kslot = (FD_Y(d_nxt) >> 3) + (lane >> 5);
            nxt.a = *(const uint4*)(Abuf + abase + ((kslot ^ axor) << 4));
        }
        d_cur = d_nxt;
        d_nxt = d_nxt_n;
        ++c;
    };
#pragma unroll 1
    while (c < nch) {   // (an odd count runs one phantom step: its MFMAs land in dead accumulators)
        step(f0, f1);
        step(f1, f0);
    }
    while (save_i < 8) save_piece();
    if (MODE == 0 && p.logits) {
        // lin_out on the rectified H3 tile still resident in the A buffer (every consumer is past the last epilogue's second
        // barrier).  w_out (fp32, <= 8 KiB) is first copied into the now idle ring -- 512 threads re-reading their slices from
        // global memory cost ~8 us per workgroup -- then 8 threads per row take 64 columns each and a butterfly adds the partials.
        float* wl = (float*)(lds + F_ABUF);
        if (tid < p.d_out * (SCENERF_D_HIDDEN / 4)) *(float4*)(wl + tid * 4) = *(const float4*)(p.w_out + tid * 4);
        asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // (the producer waves have left or are leaving: ended waves do not count)
        const int row = tid >> 3, part = tid & 7;
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

// chunk descriptors for the 32 possible scale masks, built once per (device, segment layout) and kept on that device
static SrfDescCache g_table;

int srf_desc_cache_get(SrfDescCache& C, const scenerf_cfg* cfg, hipStream_t s, int (*build)(const scenerf_cfg*, std::vector<int>&),
                       const int** desc) {
    const int dev = srf_device();
    SRF_CHECK(dev >= 0 && dev < SRF_MAX_DEVICES, "no current HIP device");
    std::lock_guard<std::mutex> lk(C.mu);
    SrfDescCache::Slot& S = C.slot[dev];
    for (int k = 0; k < SRF_DESC_LAYOUTS; ++k) {
        bool same = S.d_desc[k] != nullptr;
        for (int i = 0; i < 5; ++i) same = same && S.seg_len[k][i] == cfg->map_C[i];
        if (same) { *desc = S.d_desc[k]; return 0; }
    }
    // a new layout: built on the host and uploaded SYNCHRONOUSLY -- the table is published to every stream of the device at once, so
    // it must be complete before anyone can see the pointer (an asynchronous upload on the caller's stream ordered only that stream).
    // This is first-use setup: scenerf_hip_prepare does it before any capture; inside a hipGraph capture it fails loudly instead.
    // A small set of layouts is kept per device (alternating map layouts do not reallocate); a replaced table is NOT freed -- launches
    // of another stream may still read it -- which bounds the leak to one table (~90 KB) per replacement beyond the set.
    (void)s;
    std::vector<int> tab;
    if (int e = build(cfg, tab)) return e;
    int* d = nullptr;
    SRF_HIP(hipMalloc((void**)&d, tab.size() * sizeof(int)));
    SRF_HIP(hipMemcpy(d, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    const int k = S.next;
    S.next = (S.next + 1) % SRF_DESC_LAYOUTS;
    S.d_desc[k] = d;
    for (int i = 0; i < 5; ++i) S.seg_len[k][i] = cfg->map_C[i];
    *desc = d;
    return 0;
}


// host-only: the descriptor sets of the ring kernels (sets 0..31: forward per scale mask ; set 32: backward chain)
int fused_table_build(const scenerf_cfg* cfg, std::vector<int>& tab) {
    tab.assign((size_t)33 * F_MAXCH, 0);
    int seg_off[5], off = 0;
    for (int i = 0; i < 5; ++i) { seg_off[i] = off; off += cfg->map_C[i]; }
    SRF_CHECK(off == SCENERF_D_LATENT, "fused mlp: map channels do not add up to the latent width");
    // first w_stream block of each layer (order: w_h[0], w_fc0[0], w_h[1], w_fc0[1], w_h[2], w_fc0[2], w_h[3])
    const int layer_k[7] = {3 * SCENERF_D_XENC + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN,
                            SCENERF_D_HIDDEN + SCENERF_D_LATENT, SCENERF_D_HIDDEN, SCENERF_D_HIDDEN};
    int layer_block0[7], nb = 0;
    for (int i = 0; i < 7; ++i) { layer_block0[i] = nb; nb += layer_k[i] / F_BK; }
    SRF_CHECK(nb < 1024, "fused mlp: w_stream block index does not fit the descriptor");
    for (int mask = 0; mask < 32; ++mask) {
        int* ch = tab.data() + (size_t)mask * F_MAXCH + 1;   // entry 0 is the header
        int n = 0;
        bool ok = true;
        auto seg = [&](int layer, int src, int a0, int w0, int len) {
            if (len % F_BK || a0 % F_BK) ok = false;
            for (int k = 0; k + F_BK <= len; k += F_BK) {
                if (n >= F_MAXCH - 10) { ok = false; return; }
                ch[n] = (layer_block0[layer] + (w0 + k) / F_BK) | (((a0 + k) / F_BK) << 10) | (src << 18) | (layer << 20) | ((n % F_NST) << 25);
                ++n;
            }
        };
        auto zsegs = [&](int layer, int wbase) {
            for (int i = 0; i < 5; ++i) {
                if ((mask >> i) & 1) seg(layer, 2, seg_off[i], wbase, cfg->map_C[i]);
                wbase += cfg->map_C[i];
            }
        };
        // layer 0: [x_hi | x_lo | x_hi | z] ; layers 1,3,5: fc_0 ; layers 2,4: [relu(n) | z] ; layer 6: relu(n)
        seg(0, 1, 0, 0, 3 * SCENERF_D_XENC);
        zsegs(0, 3 * SCENERF_D_XENC);
        for (int b = 0; b < 3; ++b) {
            seg(1 + 2 * b, 0, 0, 0, SCENERF_D_HIDDEN);
            seg(2 + 2 * b, 0, 0, 0, SCENERF_D_HIDDEN);
            if (b < 2) zsegs(2 + 2 * b, SCENERF_D_HIDDEN);
        }
        SRF_CHECK(ok, "fused mlp: segment lengths must be multiples of 16 and fit the descriptor table");
        for (int i = 0; i < n; ++i) {
            if (i + 1 == n || FD_LAYER(ch[i + 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 23;
            if (i == 0 || FD_LAYER(ch[i - 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 24;
        }
        for (int i = n; i < n + 8; ++i) ch[i] = (i % F_NST) << 25;   // padding: in-bounds no-ops
        ch[-1] = n;
    }
    {   // backward chain: 6 layers of 32 chunks, operand always the resident A buffer, blocks after the forward ones
        int* ch = tab.data() + (size_t)32 * F_MAXCH + 1;
        int n = 0;
        for (int l = 0; l < 6; ++l)
            for (int i = 0; i < SCENERF_D_HIDDEN / F_BK; ++i, ++n)
                ch[n] = (nb + l * (SCENERF_D_HIDDEN / F_BK) + i) | (i << 10) | (l << 20) | ((n % F_NST) << 25) |
                        (i + 1 == SCENERF_D_HIDDEN / F_BK ? 1 << 23 : 0) | (i == 0 ? 1 << 24 : 0);
        for (int i = n; i < n + 8; ++i) ch[i] = (i % F_NST) << 25;
        ch[-1] = n;
        SRF_CHECK(nb + 6 * (SCENERF_D_HIDDEN / F_BK) == SCENERF_W_STREAM_BLOCKS, "fused mlp: w_stream block count");
    }
    return 0;
}

static int fused_table_get(const scenerf_cfg* cfg, hipStream_t s, const int** desc) {
    return srf_desc_cache_get(g_table, cfg, s, fused_table_build, desc);
}

static int fused_attrs() {
    SRF_ONCE_PER_DEVICE(
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS));
        SRF_HIP(hipFuncSetAttribute((const void*)mlp_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, F_LDS_BWD)));
    return 0;
}

int fused_prepare(const scenerf_cfg* cfg, hipStream_t s) {
    if (int e = fused_attrs()) return e;
    const int* d = nullptr;
    return fused_table_get(cfg, s, &d);
}

int launch_mlp_fwd_fused(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                         const scenerf_mlp_acts* a, hipStream_t s) {
    if (int e = fused_attrs()) return e;
    FusedArgs p = {};
    const int H = SCENERF_D_HIDDEN;
    const size_t sign_layer = (size_t)cdiv(M, SCENERF_TILE_ROWS) * SCENERF_TILE_ROWS * 64;   // bytes per layer of a->sign_bits
    auto sign = [&](int l) { return a->sign_bits ? a->sign_bits + l * sign_layer : nullptr; };
    p.layer[0] = {w->b_h[0], a->H[0], sign(0), 0, H};
    for (int b = 0; b < 3; ++b) {
        p.layer[1 + 2 * b] = {w->b_fc0[b], a->Nn[b], sign(1 + 2 * b), 1, H};
        p.layer[2 + 2 * b] = {w->b_h[b + 1], a->H[b + 1], sign(2 + 2 * b), 2, H};   // (H3's bits: wide.hip's backward makes lin_out's input
    }                                                                                 //  gradient from them, whichever kernel ran the forward)
    p.Wst = w->w_stream;
    p.X3 = a->h0pre;
    p.Z = Z;
    p.tile_mask = tile_mask;
    if (int e = fused_table_get(cfg, s, &p.desc)) return e;
    p.M = M;
    p.w_out = w->w_out;
    p.b_out = w->b_out;
    p.logits = a->logits;
    p.d_out = w->d_out;
    // FLOPs actually issued (profile mode only; synchronises to read the scale-activity mask): a 128-row tile skips the K
    // segments of the scales it does not touch in the three lin_z products
    double flops = 0;
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
    mlp_fused_kernel<0><<<cdiv(M, F_BM), F_THREADS, F_LDS, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_fused_kernel<0>");
    return 0;
}

// Backward dgrad chain: dH (act [M][2048], column block 3 = incoming gradient, written by lin_out's backward) -> column blocks
// 2, 1, 0 and dN [3][M][512].  Layer order: (fc_1.2)^T, (fc_0.2)^T, (fc_1.1)^T, (fc_0.1)^T, (fc_1.0)^T, (fc_0.0)^T -- the w_stream
// blocks after the forward operands.  Descriptor table: entry set 32 of the device table.
int launch_mlp_bwd_fused(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, int M, const scenerf_mlp_acts* a, void* dH, void* dN,
                         hipStream_t s) {
    if (int e = fused_attrs()) return e;
    FusedArgs p = {};
    const int H = SCENERF_D_HIDDEN;
    const size_t sign_layer = (size_t)cdiv(M, SCENERF_TILE_ROWS) * SCENERF_TILE_ROWS * 64;
    SRF_CHECK(a->sign_bits, "fused backward: the sign bits of the fused forward are missing");
    for (int b = 2; b >= 0; --b) {
        const int l = 2 * (2 - b);
        // forward layer order of the sign bits: H0, N0, H1, N1, H2, N2, H3
        p.layer[l] = {nullptr, (char*)dN + (size_t)b * M * H * 2, a->sign_bits + (size_t)(2 * b + 1) * sign_layer, 1, H};   // dN_b = (dH_{b+1} W1_b) * [N_b > 0]
        p.layer[l + 1] = {nullptr, (char*)dH + (size_t)b * H * 2, a->sign_bits + (size_t)(2 * b) * sign_layer, 2, 4 * H};   // dH_b = dH_{b+1} + (dN_b W0_b) * [H_b > 0]
    }
    p.Wst = w->w_stream;
    p.dH3 = (const char*)dH + (size_t)3 * H * 2;
    p.dH_ld = 4 * H;
    const int* desc = nullptr;
    if (int e = fused_table_get(cfg, s, &desc)) return e;
    p.desc = desc + 32 * F_MAXCH;
    p.M = M;
    SrfLaunchScope ps(s, w->d_out == 2 ? "mlp_bwd_fused/g" : "mlp_bwd_fused", 2.0 * M * 512.0 * 6.0 * 512.0, 0);
    mlp_fused_kernel<1><<<cdiv(M, F_BM), F_THREADS, F_LDS_BWD, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_fused_kernel<1>");
    return 0;
}
