// The 16-bit 256 x 256 tile on FOUR waves with the 32 x 32 x 16 matrix instruction (operator form: NT, 16-bit out).
//   * one wave per SIMD, 128 x 128 wave tiles: a third fewer LDS-read bytes per flop than the 8-wave build's 64 x 128;
//   * the wave's 256 accumulators live in AGPRs -- the 16-bit kernels accumulate inside the matrix instruction, nothing promotes
//     them through the vector ALU (this translation unit is compiled WITHOUT -amdgpu-mfma-vgpr-form);
//   * 32 x 32 x 16: half the matrix instructions of the 16 x 16 x 32 form (32 cycles each), so the one wave of a SIMD has twice the
//     time per instruction to slip its fragment reads, refill DMAs and barrier in between;
//   * same LDS image, DMA and continuous schedule as gemm_b16_kernel.hpp's PP = 2 loop: k step = 64 elements, stage = parity of
//     the step, ONE barrier per step in front of the last n-tile, A fragments reloaded in place under the last n-tile.
// Fragment geometry (v_mfma_f32_32x32x16, D = A B + C, lane l supplies A[l % 32][8 (l / 32) ..+7] and B[8 (l / 32) ..+7][l % 32]; D register v of
// lane l is row 8 (v / 4) + 4 (l / 32) + v % 4, column l % 32): the y tile is the A operand (rows = n), the x tile the B operand (columns = m).
// y rows are read PERMUTED -- lane i < 16 takes row 8 (i / 4) + i % 4, lane i >= 16 row 8 ((i - 16) / 4) + i % 4 + 4 -- which is the
// row set the B image's swizzle is conflict-free on, and leaves every lane with two runs of 8 consecutive n (16-byte stores).
#pragma once
#include "gemm_b16_kernel.hpp"

namespace dga {

typedef float v16f __attribute__((ext_vector_type(16)));

template <bool BF16>
__device__ __forceinline__ v16f mfma32_b16(v4i a, v4i b, v16f c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf16, a), __builtin_bit_cast(v8bf16, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8f16, a), __builtin_bit_cast(v8f16, b), c, 0, 0, 0);
}

template <bool BF16>
__device__ __forceinline__ v4i pack8_b16(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7)
{
    if constexpr (BF16) {
        return v4i{__builtin_bit_cast(int, __builtin_convertvector(v2f{a0, a1}, v2bf)), __builtin_bit_cast(int, __builtin_convertvector(v2f{a2, a3}, v2bf)),
                   __builtin_bit_cast(int, __builtin_convertvector(v2f{a4, a5}, v2bf)), __builtin_bit_cast(int, __builtin_convertvector(v2f{a6, a7}, v2bf))};
    } else {
        typedef _Float16 v2h __attribute__((ext_vector_type(2)));
        return v4i{__builtin_bit_cast(int, v2h{(_Float16)a0, (_Float16)a1}), __builtin_bit_cast(int, v2h{(_Float16)a2, (_Float16)a3}),
                   __builtin_bit_cast(int, v2h{(_Float16)a4, (_Float16)a5}), __builtin_bit_cast(int, v2h{(_Float16)a6, (_Float16)a7})};
    }
}

// whole raster, batch 1, no split-K (p.launch_tiles / tail fields unused); p.z16 = out
template <bool BF16>
__global__ void __launch_bounds__(256) gemm_b16_w4_kernel(const B16Params p)
{
    using Cfg = GemmCfg<256, 256, 2, 2, 2>;
    constexpr int BM = 256, BN = 256, DNT = Cfg::DNT, STAGE = Cfg::A_BYTES + Cfg::B_BYTES, NL = Cfg::A_ITERS + Cfg::B_ITERS;
    constexpr int TM = 4, TN = 4, KC = 4;                       // 32-row m-tiles, 32-column n-tiles, 16-element k chunks of a step
    static_assert(DNT == 256 && Cfg::A_ITERS == 8 && Cfg::B_ITERS == 8, "four DMA waves");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    int tile;
    {
        const int nwg = gridDim.x, bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tm, tn;
    {
        const int gm = p.raster_group, per = gm * p.tiles_n, band = tile / per, first = band * gm;
        const int rows = min(p.tiles_m - first, gm), loc = tile - band * per;
        tm = first + loc % rows;
        tn = loc / rows;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const uint8_t *X = (const uint8_t *)p.x, *Y = (const uint8_t *)p.yt;
    const int64_t ldxb = p.ldx * 2, ldyb = p.ldy * 2;

    const int a_col = ((tid & 7) ^ swz_a(tid >> 3)) * 16, b_col = ((tid & 7) ^ swz_b(tid >> 3)) * 16;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::A_ITERS; ++it) a_voff[it] = (uint32_t)min((it * DNT + tid) >> 3, p.m - 1 - m0) * (uint32_t)ldxb + a_col;
#pragma unroll
    for (int it = 0; it < Cfg::B_ITERS; ++it) b_voff[it] = (uint32_t)min((it * DNT + tid) >> 3, p.n - 1 - n0) * (uint32_t)ldyb + b_col;
    const v4i a_rsrc = make_rsrc(X + (int64_t)m0 * ldxb, (int64_t)(p.m - m0) * ldxb);
    const v4i b_rsrc = make_rsrc(Y + (int64_t)n0 * ldyb, (int64_t)(p.n - n0) * ldyb);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    auto issue_one = [&](int idx, int stage, int ks) {
        const uint32_t sa = lds0 + stage * STAGE + wave * 1024;
        if (idx < Cfg::A_ITERS) dma16(a_voff[idx], a_rsrc, (uint32_t)(ks * 128), sa + idx * DNT * 16);
        else dma16(b_voff[idx - Cfg::A_ITERS], b_rsrc, (uint32_t)(ks * 128), sa + Cfg::A_BYTES + (idx - Cfg::A_ITERS) * DNT * 16);
    };

    const int l32 = lane & 31, h = lane >> 5;
    const int a_row = wm * 128 + l32;                            // + 32 mt
    const int b_perm = l32 < 16 ? 8 * (l32 >> 2) + (l32 & 3) : 8 * ((l32 - 16) >> 2) + (l32 & 3) + 4;
    const int b_row = wn * 128 + b_perm;                         // + 32 nt
    int a_off[KC], b_off[KC];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        a_off[kc] = a_row * 128 + (((2 * kc + h) ^ swz_a(a_row)) * 16);
        b_off[kc] = Cfg::A_BYTES + b_row * 128 + (((2 * kc + h) ^ swz_b(b_row)) * 16);
    }

    v16f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

    const int KS = p.k / 64;
    constexpr int STEPS = TN * KC * TM, SB = (TN - 1) * KC * TM;
    // (all sixteen refills right behind the barrier -- a whole k step to land -- crowd the last n-tile, which also carries the sixteen
    //  in-place A reloads: 106 against 102 us at 4096^3; half there, half on the head of the next step, as in the 8-wave loop)
    constexpr int TAIL_DMA = NL / 2, HEAD_STEPS = (STEPS * 9) / 32;   // (denser -- 8 steps -- or all sixteen behind the barrier: 106-107 us; over 32 steps: 103)
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    v4i af[TM][KC], bfr[2][KC];
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) issue_one(idx, 0, 0);
#pragma unroll
    for (int idx = 0; idx < TAIL_DMA; ++idx) issue_one(idx, 1, 1);
    wait_vmcnt<TAIL_DMA>();
    barrier();
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) bfr[0][kc] = *(const v4i *)(smem + b_off[kc]);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int kc = 0; kc < KC; ++kc) af[mt][kc] = *(const v4i *)(smem + a_off[kc] + mt * 4096);
#ifdef DGA_W4_STAMPS
    const uint64_t t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    uint64_t wait_cycles = 0;
#endif
    for (int ks = 0; ks < KS; ++ks) {
        const int par = ks & 1;
        const uint8_t *st = smem + par * STAGE;
        const uint8_t *sn = smem + (par ^ 1) * STAGE;
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            const int nt = i / (KC * TM), kc = (i / TM) % KC, mt = i % TM;
            if (i == SB) {
#ifdef DGA_W4_STAMPS
                const uint64_t w0 = __builtin_readcyclecounter();
#endif
                wait_vmcnt<0>();
                barrier();
#ifdef DGA_W4_STAMPS
                wait_cycles += __builtin_readcyclecounter() - w0;
#endif
            }
            acc[mt][nt] = mfma32_b16<BF16>(bfr[nt & 1][kc], af[mt][kc], acc[mt][nt]);
            __builtin_amdgcn_sched_barrier(0);
            if (i < HEAD_STEPS) {   // head part of k step ks + 1's refill (stage par ^ 1 is free since the previous barrier)
#pragma unroll
                for (int j = (i * (NL - TAIL_DMA)) / HEAD_STEPS; j < ((i + 1) * (NL - TAIL_DMA)) / HEAD_STEPS; ++j)
                    issue_one(TAIL_DMA + j, par ^ 1, ks + 1);
            }
            if (i > SB && i <= SB + TAIL_DMA) issue_one(i - SB - 1, par, ks + 2);  // tail part of ks + 2 into this stage
            if (kc == 0 && mt == 0) {  // the next n-tile's y fragments (wraps into the next k step)
                const uint8_t *src = nt + 1 < TN ? st : sn;
                const int nn = nt + 1 < TN ? nt + 1 : 0;
#pragma unroll
                for (int c = 0; c < KC; ++c) bfr[(nt + 1) & 1][c] = *(const v4i *)(src + b_off[c] + nn * 4096);
            }
            if (nt == TN - 1) af[mt][kc] = *(const v4i *)(sn + a_off[kc] + mt * 4096);  // in-place reload for the next k step
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    wait_vmcnt<0>();
#ifdef DGA_W4_STAMPS
    if (p.partial && lane == 0) {
        const uint64_t t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
        uint64_t *o = (uint64_t *)p.partial + ((size_t)blockIdx.x * 4 + wave) * 3;
        o[0] = t1 - t0; o[1] = r1 - r0; o[2] = wait_cycles;
    }
#endif

    // epilogue: lane (l32, h) owns row m = 32 mt + l32 of its wave's rows; of n-tile nt the columns 8 h + [0, 8) (registers 0-3, 8-11)
    // and 16 + 8 h + [0, 8) (registers 4-7, 12-15)
    uint16_t *Z16 = p.z16;
    const bool v16_ok = ((p.n & 7) == 0) && ((((uintptr_t)Z16) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = m0 + wm * 128 + mt * 32 + l32;
        if (m >= p.m) continue;
        uint16_t *zr = Z16 + (int64_t)m * p.n;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const v16f a = acc[mt][nt];
#pragma unroll
            for (int run = 0; run < 2; ++run) {
                const int n = n0 + wn * 128 + nt * 32 + 16 * run + 8 * h;
                const v4i pk = run == 0 ? pack8_b16<BF16>(a[0], a[1], a[2], a[3], a[8], a[9], a[10], a[11])
                                        : pack8_b16<BF16>(a[4], a[5], a[6], a[7], a[12], a[13], a[14], a[15]);
                if (v16_ok && n + 8 <= p.n) {
                    *(v4i *)(zr + n) = pk;
                } else {
                    const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (n + q < p.n) zr[n + q] = e[q];
                }
            }
        }
    }
}

}  // namespace dga
