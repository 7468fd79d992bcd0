// Register-streamed fused ResnetFC forward for gfx950 (bf16 operands): the whole 7-GEMM trunk (lin_in + lin_z.0, three residual
// blocks fc_0 / fc_1 + lin_z.b) and lin_out for a block of 64 rows in ONE kernel.  reference scenerf/models/resnetfc.py:133-164.
//
// Same arithmetic, data layouts and HBM traffic as the LDS-ring kernel in fused.hip (results are bit-identical), different plumbing:
//   * 8 waves and NO weight ring in LDS: wave w owns output columns [64 w, 64 w + 64) for all 64 rows (2 x 2 MFMA 32x32x16 tiles)
//     and streams exactly ITS slice of w_stream (two coalesced 1-KiB global_load_dwordx4 per 16-wide K chunk) straight into a VGPR
//     ring, four steps deep -- nothing is shared between waves, so a hidden layer's K loop runs without a single barrier;
//   * a step covers a PAIR of chunks (K = 32): what limits a 64-row block is not memory but instruction issue -- a wave issues at
//     most one instruction per 4 cycles, and a step's bookkeeping (descriptor window, branches, addresses: ~45 instructions)
//     against 4 MFMAs left the matrix pipe 40 % busy whatever the memory system did; 8 MFMAs per step halve that overhead;
//   * the loads are plain C++ loads: hipcc counts vmcnt for them by itself once a scheduling barrier per step stops it from
//     sinking them to their first use.  Therefore NO other load kind may sit in the loop (an LDS-DMA makes hipcc wait vmcnt(0)):
//     the streamed activation operand of the lin_in / lin_z segments (X3 / Z rows, shared by all waves) is register-staged -- in
//     step 0 of every group of four steps wave pair q writes chunk pair c + 1 + q to one of eight 4-KiB LDS stages and fetches chunk
//     pair c + 5 + q; only streamed steps start with a barrier; the activation / sign-bit stores are inline asm (the compiler
//     must not see them: a pending store would turn every counted wait into vmcnt(0); hidden stores only make a counted wait more
//     conservative);
//   * the first eight steps of every layer that starts with the resident operand run as a "fast run": fragment addresses from eight
//     precomputed registers + immediate offsets, the weight block from one v_readlane -- ~16 non-MFMA instructions per 8 MFMAs;
//   * layer ends fall on multiples of four steps (host-padded with no-op pairs; odd segments end in a half no-op pair) so that the
//     ring slot is static in the 4x unrolled loop and the epilogue has ONE site; biases of all layers, w_out and the descriptors sit
//     in LDS from the start (121 KiB in all).
#include "fused.h"
#include <type_traits>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4_f;

#define G_THREADS 512
#define G_D 4                                  // weight ring depth in steps (= steps per group of the unrolled loop)
#define G_NW 12                                // descriptor window: steps c .. c + 11 (staging looks 8 ahead; 12 keeps the four that
                                               // join it per group inside one 64-entry block of the table)
#define G_NSTG 8                               // streamed-operand stages
#define G_STGB (2 * F_A2STG)                   // 4 KiB: both chunks of a step
#define G_STG F_ABUF                           // 65536
#define G_BIAS (G_STG + G_NSTG * G_STGB)       // 98304: 7 layers x 2 KiB
#define G_WOUT (G_BIAS + 7 * 2048)             // 112640: w_out, <= 8 KiB
#define G_TAB (G_WOUT + 8192)                  // 120832: this tile mask's step descriptors (+ read slack)
#define G_LDS (G_TAB + 768 * 4)                // 123904
// One descriptor per step = pair of consecutive 16-wide K chunks: fields as in fused.h for the FIRST chunk (w_stream block, A column,
// src, layer, end) -- the second chunk is block + 1 / column + 16 -- except [25] = the second chunk is a no-op (odd segment: staged as zeros),
// [26:28] = stage (step mod 8), [29] = the whole step is a no-op (padding; the loads still happen, from valid addresses)
#define GD_SKIP2(d) (((d) >> 25) & 1)
#define GD_STAGE(d) (((d) >> 26) & 7)
#define GD_SKIPALL(d) (((d) >> 29) & 1)
#define GD_FAST(d) (((d) >> 30) & 1)           // first step of a layer whose first G_FASTN steps may run as one tight loop (below)
#define G_FASTN 8                              // steps per fast run (16 chunks)

__device__ static inline void g_store16(void* p, uint4 v) {
    const u32x4_f t = {v.x, v.y, v.z, v.w};
    // (s_nop: a 16-byte store reads its data registers after issue, and the compiler -- which does not know this statement is a
    //  store -- may overwrite them with the very next instruction; without the wait states the 128-row experiment stored garbage)
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
}
__device__ static inline void g_store1(void* p, uint32_t v) { asm volatile("global_store_byte %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

__device__ static inline uint32_t g_pk_min_u16(uint32_t a, uint32_t b) {   // v_pk_min_u16, by name (see wide.hip: the compiler's own form of
    uint32_t r;                                                                   // min(x, 1) is compare + select per half)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "s"(b));
    return r;
}

__global__ __launch_bounds__(G_THREADS) void mlp_stream_kernel(FusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* const Abuf = lds;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wvu = __builtin_amdgcn_readfirstlane(wv);
    const int m0 = blockIdx.x * F_BM;
    const unsigned mask = __builtin_amdgcn_readfirstlane((unsigned)p.tile_mask[m0 / SCENERF_TILE_ROWS] & 31u);
    desc_ptr D = (desc_ptr)(uintptr_t)(p.desc + mask * F_MAXCH);
    const int nch = D[0];   // header: number of steps (a multiple of 4); the descriptors follow, zero-padded
    ++D;

    // ---- one-time LDS contents: the biases of all seven layers, w_out
    for (int i = tid; i < 7 * 128; i += G_THREADS)
        *(float4*)(lds + G_BIAS + i * 16) = *(const float4*)(p.layer[i >> 7].bias + (i & 127) * 4);
    if (p.logits && tid < p.d_out * (SCENERF_D_HIDDEN / 4)) *(float4*)(lds + G_WOUT + tid * 16) = *(const float4*)(p.w_out + tid * 4);
    // the descriptors too: a scalar load per step shares lgkmcnt with the fragment reads and returns out of order, so every step
    // would start by waiting out its latency; from LDS, 64 at a time into one VGPR + v_readlane (no measurable difference in the end)
    int* const tab = (int*)(lds + G_TAB);
    for (int i = tid; i < nch + G_NW + 8; i += G_THREADS) tab[i] = D[i];

    // ---- weights: lane's 16 bytes of tile j of a 16-KiB w_stream block ([512 rows n][32 B], halves swapped when (n >> 3) & 1)
    const uint4* const Wb = (const uint4*)p.Wst;
    const int woff = (wvu * 64 + (lane & 31)) * 2 + ((lane >> 5) ^ ((lane >> 3) & 1));   // + 64 for tile 1
    // ---- streamed operand (X3 / Z rows): the pair's two waves cover rows 0..31 / 32..63; lane -> row lane / 2, physical slot lane & 1
    const int pr = wvu >> 1;
    const int gm_a = min(m0 + 32 * (wvu & 1) + (lane >> 1), p.M - 1);
    const int pls = ((lane & 1) ^ ((lane >> 4) & 1)) << 4;
    const unsigned ox3 = (unsigned)gm_a * (3 * SCENERF_D_XENC * 2) + pls;   // < 4 GiB: M * 4960 B fits 32 bits up to 865k rows
    const unsigned oz = (unsigned)gm_a * (SCENERF_D_LATENT * 2) + pls;
    auto s_load = [&](const int d, uint4& v0, uint4& v1) __attribute__((always_inline)) {   // (wave-uniform d, src != 0) both chunks of a step
        const char* base = FD_SRC(d) == 1 ? (const char*)p.X3 : (const char*)p.Z;
        const unsigned o = (FD_SRC(d) == 1 ? ox3 : oz) + (unsigned)FD_Y(d) * 2;
        v0 = *(const uint4*)(base + o);
        v1 = *(const uint4*)(base + o + (GD_SKIP2(d) ? 0u : 32u));   // (a no-op second chunk re-reads the first: never past the buffer ...
        if (GD_SKIP2(d)) v1 = uint4{0, 0, 0, 0};                     //  ... and is staged as zeros: its MFMAs add exactly nothing, no branch)
    };
    auto s_write = [&](const int d, const uint4 v0, const uint4 v1) __attribute__((always_inline)) {
        char* st = lds + G_STG + GD_STAGE(d) * G_STGB + (wvu & 1) * 1024 + lane * 16;
        *(uint4*)st = v0;
        *(uint4*)(st + F_A2STG) = v1;
    };

    // ---- fragments.  Transposed accumulator tile (i, j): lane holds activation row m = 32 i + (lane & 31) and outputs
    // n = 64 w + 32 j + 8 q + 4 (lane >> 5) + e in register 4 q + e.  hp: the residual stream as packed bf16 pairs.
    f32x16_f acc[2][2];
    uint32_t hp[2][2][8];
    const int arow = (lane & 31) * F_AROW;
    const int axor = lane & 15;
    const int offA2 = (lane & 31) * 32 + (((lane >> 5) ^ ((lane >> 3) & 1)) << 4);
    auto init_acc = [&](const int layer) __attribute__((always_inline)) {
        const char* bb = lds + G_BIAS + layer * 2048 + (wvu * 64 + 4 * (lane >> 5)) * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b = *(const float4*)(bb + (j * 32 + q * 8) * 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) { acc[i][j][4 * q] = b.x; acc[i][j][4 * q + 1] = b.y; acc[i][j][4 * q + 2] = b.z; acc[i][j][4 * q + 3] = b.w; }
            }
    };

    // ---- layer output -> HBM: the A buffer of the finished layer is streamed out one 16-byte piece per thread per step of the
    // NEXT layer (8 pieces), stores and sign bytes by inline asm
    char* save_ptr = nullptr;
    uint8_t* sign_ptr = nullptr;
    int save_ld2 = 0;
    int save_i = 8;
    auto save_piece = [&]() __attribute__((always_inline)) {   // rows 8 i .. 8 i + 7: thread t moves slot t & 63 of row 8 i + t / 64
        const int row = 8 * save_i + (tid >> 6), slot = tid & 63;
        if (save_ptr && m0 + row < p.M) {
            const uint4 v = *(const uint4*)(Abuf + row * F_AROW + ((slot ^ (row & 15)) << 4));
            g_store16(save_ptr + (size_t)(m0 + row) * save_ld2 + slot * 16, v);
            if (sign_ptr) {   // 8 sign bits per piece (rectified values: positive == non-zero), see fused.hip
                uint32_t u = g_pk_min_u16(v.x, 0x00010001u);
                u |= g_pk_min_u16(v.y, 0x00010001u) << 2;
                u |= g_pk_min_u16(v.z, 0x00010001u) << 4;
                u |= g_pk_min_u16(v.w, 0x00010001u) << 6;
                g_store1(sign_ptr + (size_t)(m0 + row) * 64 + slot, (u | (u >> 15)) & 0xffu);
            }
        }
        ++save_i;
    };

    // ---- layer epilogue: residual in registers, rectified output -> resident A buffer
    auto epilogue = [&](const int layer) __attribute__((always_inline)) {
        const FusedLayer& L = p.layer[layer];
        while (save_i < 8) save_piece();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // every wave has finished reading the A buffer for this layer
        const bool is_res = L.kind != 1;  // residual layers: out = h + acc, h = bf16(out) ; fc_0 layers: out = acc   (bias is in acc)
        const float resf = is_res ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            int wbase = (32 * i + (lane & 31)) * F_AROW + 8 * (lane >> 5);
            asm volatile("" : "+v"(wbase));
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint32_t pk[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        const int k = 2 * q + e;
                        const float v0 = __builtin_fmaf(bf16lo(hp[i][j][k]), resf, acc[i][j][2 * k]);
                        const float v1 = __builtin_fmaf(bf16hi(hp[i][j][k]), resf, acc[i][j][2 * k + 1]);
                        pk[e] = pack_bf16x2(v0, v1);
                        hp[i][j][k] = is_res ? pk[e] : hp[i][j][k];
                    }
                    const int slot = wvu * 8 + j * 4 + q;
                    uint2 o;
                    o.x = relu_bf16x2(pk[0]);
                    o.y = relu_bf16x2(pk[1]);
                    *(uint2*)(Abuf + wbase + ((slot ^ axor) << 4)) = o;
                }
        }
        asm volatile("" ::: "memory");
        init_acc(layer < 6 ? layer + 1 : 6);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();     // A buffer complete
        save_ptr = (char*)L.save;
        sign_ptr = L.sign;
        save_ld2 = L.save_ld * 2;
        save_i = 0;
    };

    // ---- prologue: descriptor window, weight ring (chunks 0..3), streamed chunks 0..5
    // (named scalars, not an array: hipcc turns a select chain over array elements into a dynamically indexed load and the array
    //  moves to scratch memory)
    int q0 = D[0], q1 = D[1], q2 = D[2], q3 = D[3], q4 = D[4], q5 = D[5], q6 = D[6], q7 = D[7], q8 = D[8], q9 = D[9], q10 = D[10], q11 = D[11];
    // staging registers of this wave's pair: all pairs stage in step 0 of a group (wave pair q: step c + 1 + q written, step c + 5 + q
    // fetched) -- ONE load site, so hipcc sees the ring loads of four steps between the fetch and the write and keeps its wait counted
    uint4 zr0 = {0, 0, 0, 0}, zr1 = {0, 0, 0, 0};
    {   // step 0: straight into its stage (wave pair 0); steps 1..4: fetched here, written in step 0
        if (pr == 0 && FD_SRC(q0)) { uint4 t0, t1; s_load(q0, t0, t1); s_write(q0, t0, t1); }
        const int dlat = pr == 0 ? q1 : pr == 1 ? q2 : pr == 2 ? q3 : q4;
        if (FD_SRC(dlat)) s_load(dlat, zr0, zr1);
    }
    // (the ring loads come AFTER the staging loads: the first staging write then has a full ring of younger loads in front of it on
    // every path, like in the steady state, and hipcc's counted wait there does not drain the ring)
    uint4 ring[G_D][2][2];
#pragma unroll
    for (int s = 0; s < G_D; ++s) {
        const uint4* b = Wb + (size_t)FD_Z(s == 0 ? q0 : s == 1 ? q1 : s == 2 ? q2 : q3) * 1024 + woff;
        ring[s][0][0] = b[0];
        ring[s][0][1] = b[64];
        ring[s][1][0] = b[1024];
        ring[s][1][1] = b[1024 + 64];
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) hp[i][j][k] = 0u;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();   // biases, w_out, stages 0 / 1 (a full barrier: the global loads above are waited for too)
    init_acc(0);
    int dv = tab[lane];                   // descriptors 64 k .. 64 k + 63 of the block the window's head (c + G_NW) is in
    __builtin_amdgcn_sched_barrier(0);

    int c = 0;
    int dend = 0;                         // descriptor of the last step of the group just done
    auto step = [&](auto SC) __attribute__((always_inline)) {
        constexpr int S = decltype(SC)::value;
        const int idx = c + G_NW;         // joins the window at the end of the step
        if ((idx & 63) == 0) dv = tab[idx + lane];
        const int dn = __builtin_amdgcn_readlane(dv, idx & 63);
        const int d0 = q0;
        if (FD_SRC(d0)) {                 // streamed step: its stage was written at least one step ago by some wave pair
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        }
        uint4 a[2][2];                    // [chunk][row tile]
        if (FD_SRC(d0) == 0) {
            const int kslot = (FD_Y(d0) >> 3) + (lane >> 5);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i) a[k][i] = *(const uint4*)(Abuf + i * 32 * F_AROW + arow + (((kslot + 2 * k) ^ axor) << 4));
        } else {
            const char* St = lds + G_STG + GD_STAGE(d0) * G_STGB + offA2;
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i) a[k][i] = *(const uint4*)(St + k * F_A2STG + i * 1024);
        }
        if (!GD_SKIPALL(d0)) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)   // C^T tile: rows = outputs n, cols = activation rows m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_f, ring[S][0][j]), __builtin_bit_cast(bf16x8_f, a[0][i]),
                                                                        acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)   // (the second chunk of an odd segment's last pair is staged as zeros)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_f, ring[S][1][j]), __builtin_bit_cast(bf16x8_f, a[1][i]),
                                                                        acc[i][j], 0, 0, 0);
            // all four fragment reads first, then the MFMAs back to back (left alone, hipcc reuses one fragment register set and
            // exposes the LDS latency four times per step)
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
        }
        {   // refill this ring slot with step c + 4 (two consecutive blocks)
            const uint4* b = Wb + (size_t)FD_Z(q4) * 1024 + woff;
            ring[S][0][0] = b[0];
            ring[S][0][1] = b[64];
            ring[S][1][0] = b[1024];
            ring[S][1][1] = b[1024 + 64];
        }
        if (S == 0) {                     // staging: wave pair q writes step c + 1 + q (used >= 1 step later), fetches step c + 5 + q
            const int dw = pr == 0 ? q1 : pr == 1 ? q2 : pr == 2 ? q3 : q4;
            const int dl = pr == 0 ? q5 : pr == 1 ? q6 : pr == 2 ? q7 : q8;
            if (FD_SRC(dw)) s_write(dw, zr0, zr1);
            if (FD_SRC(dl)) s_load(dl, zr0, zr1);
        }
        if (save_i < 8) save_piece();
        if (S == G_D - 1) dend = d0;      // (layer ends only here: the epilogue runs after the group, at its single site)
        q0 = q1; q1 = q2; q2 = q3; q3 = q4; q4 = q5; q5 = q6; q6 = q7; q7 = q8; q8 = q9; q9 = q10; q10 = q11; q11 = dn;
        ++c;
        __builtin_amdgcn_sched_barrier(0);   // nothing moves across a step: the ring loads stay where they are written
    };
    // Fast run: the first G_FASTN steps of a layer whose K loop starts with the resident operand (every layer but the first) when no
    // staging work falls into them (host flag).  Same arithmetic as G_FASTN steps, but with the bookkeeping a step pays per pair
    // of chunks -- descriptor window, branches, address arithmetic: ~70 instructions per 8 MFMAs, which is what holds the MFMA pipe
    // at ~40 % (tools/ubench/mfma_overlap) -- reduced to the operands themselves: fragment addresses are eight precomputed
    // registers + immediate offsets (the XOR swizzle only sees the slot index mod 16), the weight block of step c + 4 comes from
    // one v_readlane, the first eight steps carry the previous layer's eight save pieces.
    int aoff[4][2];                       // [step & 3][chunk]: row base + swizzled 16-byte slot of this lane
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int k = 0; k < 2; ++k) aoff[s4][k] = arow + (((4 * s4 + 2 * k + (lane >> 5)) ^ axor) << 4);
    auto fast_run = [&]() __attribute__((always_inline)) {
        const int dvf = tab[c + lane];    // descriptors of steps c .. c + 63 (the run and the steps after it)
#pragma unroll
        for (int t = 0; t < G_FASTN; ++t) {
            uint4 a[2][2];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i) a[k][i] = *(const uint4*)(Abuf + i * 32 * F_AROW + (t >> 2) * 256 + aoff[t & 3][k]);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_f, ring[t & 3][k][j]), __builtin_bit_cast(bf16x8_f, a[k][i]),
                                                                            acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
            const uint4* b = Wb + (size_t)FD_Z(__builtin_amdgcn_readlane(dvf, t + 4)) * 1024 + woff;
            ring[t & 3][0][0] = b[0];
            ring[t & 3][0][1] = b[64];
            ring[t & 3][1][0] = b[1024];
            ring[t & 3][1][1] = b[1024 + 64];
            save_piece();                 // (a fast run starts a layer: pieces 0 .. 7 of the previous layer's output)
            __builtin_amdgcn_sched_barrier(0);
        }
        c += G_FASTN;
        // the window behind the run, and the descriptor block the next step's window head (c + G_NW) lives in
        q0 = __builtin_amdgcn_readlane(dvf, G_FASTN + 0); q1 = __builtin_amdgcn_readlane(dvf, G_FASTN + 1);
        q2 = __builtin_amdgcn_readlane(dvf, G_FASTN + 2); q3 = __builtin_amdgcn_readlane(dvf, G_FASTN + 3);
        q4 = __builtin_amdgcn_readlane(dvf, G_FASTN + 4); q5 = __builtin_amdgcn_readlane(dvf, G_FASTN + 5);
        q6 = __builtin_amdgcn_readlane(dvf, G_FASTN + 6); q7 = __builtin_amdgcn_readlane(dvf, G_FASTN + 7);
        q8 = __builtin_amdgcn_readlane(dvf, G_FASTN + 8); q9 = __builtin_amdgcn_readlane(dvf, G_FASTN + 9);
        q10 = __builtin_amdgcn_readlane(dvf, G_FASTN + 10); q11 = __builtin_amdgcn_readlane(dvf, G_FASTN + 11);
        dv = tab[((c + G_NW - 1) & ~63) + lane];
        __builtin_amdgcn_sched_barrier(0);
    };
#pragma unroll 1
    while (c < nch) {
        if (GD_FAST(q0)) fast_run();
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        step(std::integral_constant<int, 2>{});
        step(std::integral_constant<int, 3>{});
        if (FD_END(dend)) epilogue(FD_LAYER(dend));
        __builtin_amdgcn_sched_barrier(0);
    }
    while (save_i < 8) save_piece();
    if (p.logits) {
        // lin_out on the rectified H3 tile still resident in the A buffer (all waves are past the last epilogue's second barrier);
        // 8 threads per row take 64 columns each, a butterfly adds the partials
        const float* wl = (const float*)(lds + G_WOUT);
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

// chunk descriptors for the 32 possible scale masks (forward): like fused.hip's table, every layer padded to a multiple of four chunks
static SrfDescCache g_stream_table;

// host-only: the 32 descriptor sets of the streamed forward
int stream_table_build(const scenerf_cfg* cfg, std::vector<int>& tab) {
    tab.assign((size_t)32 * F_MAXCH, 0);
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
        auto seg = [&](int layer, int src, int a0, int w0, int len) {   // one descriptor per PAIR of chunks of the segment
            if (len % F_BK || a0 % F_BK || w0 % F_BK) ok = false;
            const int nchunk = len / F_BK;
            for (int k = 0; k < nchunk; k += 2) {
                if (n >= F_MAXCH - 20) { ok = false; return; }
                ch[n] = (layer_block0[layer] + w0 / F_BK + k) | ((a0 / F_BK + k) << 10) | (src << 18) | (layer << 20) |
                        (k + 1 >= nchunk ? 1 << 25 : 0) | ((n % G_NSTG) << 26);
                ++n;
            }
        };
        auto zsegs = [&](int layer, int wbase) {
            for (int i = 0; i < 5; ++i) {
                if ((mask >> i) & 1) seg(layer, 2, seg_off[i], wbase, cfg->map_C[i]);
                wbase += cfg->map_C[i];
            }
        };
        auto pad = [&](int layer) {   // no-op steps up to a multiple of four: resident operand, blocks 0 / 1, MFMAs skipped
            while (n % G_D) { ch[n] = (layer << 20) | (1 << 29) | ((n % G_NSTG) << 26); ++n; }
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
        SRF_CHECK(ok && n % G_D == 0, "stream mlp: segment lengths must be multiples of 16 and fit the descriptor table");
        for (int i = 0; i < n; ++i) {
            if (i + 1 == n || FD_LAYER(ch[i + 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 23;
            if (i == 0 || FD_LAYER(ch[i - 1]) != FD_LAYER(ch[i])) ch[i] |= 1 << 24;
        }
        // fast runs: the first G_FASTN steps of a layer, all from the resident operand and real, the layer longer than the run
        // (so no layer end inside), and no staging work in the run's groups (the group at step g writes steps g+1 .. g+4 and
        // fetches steps g+5 .. g+8: steps f+1 .. f+G_FASTN+4 must not be streamed)
        for (int f = 0; f + G_FASTN < n; f += 4) {
            if (!FD_BEGIN(ch[f])) continue;
            bool fast = true;
            for (int i = 0; i < G_FASTN; ++i) fast = fast && FD_SRC(ch[f + i]) == 0 && !GD_SKIPALL(ch[f + i]) && !FD_END(ch[f + i]);
            for (int i = 1; i <= G_FASTN + 4; ++i) fast = fast && FD_SRC(ch[f + i]) == 0;
            if (fast) ch[f] |= 1 << 30;
        }
        ch[-1] = n;   // entries n .. n + 15 stay zero: prefetches past the end read block 0 and are never used
    }
    return 0;
}

static int stream_table_get(const scenerf_cfg* cfg, hipStream_t s, const int** desc) {
    return srf_desc_cache_get(g_stream_table, cfg, s, stream_table_build, desc);
}

static int stream_attrs() {
    SRF_ONCE_PER_DEVICE(SRF_HIP(hipFuncSetAttribute((const void*)mlp_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS)));
    return 0;
}

int stream_prepare(const scenerf_cfg* cfg, hipStream_t s) {
    if (int e = stream_attrs()) return e;
    const int* d = nullptr;
    return stream_table_get(cfg, s, &d);
}

int launch_mlp_fwd_stream(const scenerf_cfg* cfg, const scenerf_mlp_weights* w, const void* Z, const uint8_t* tile_mask, int M,
                          const scenerf_mlp_acts* a, hipStream_t s) {
    if (int e = stream_attrs()) return e;
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
    mlp_stream_kernel<<<cdiv(M, F_BM), G_THREADS, G_LDS, s>>>(p);
    SRF_LAUNCH_CHECK("mlp_stream_kernel");
    return 0;
}
