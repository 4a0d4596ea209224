// One-launch split-K INSIDE a workgroup for short-M ("decode") problems: the 8 waves of a workgroup take the 8 K slices of ONE
// output tile, each wave streams its slice of A and B straight from global memory into MFMA fragment registers (nothing is
// shared between the waves while they stream: no LDS, no barrier, no DMA ring), and the eight fp32 partial tiles meet in LDS
// behind one barrier -- no slab in HBM, no ticket, no combine launch.
//
// Reference counterparts: the fused reduce of the Stream-K kernel
// (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_streamk_matmul_kernel.h:92-107), the selection rule
// (op_host/op_tiling/select_kernel.cpp:303-331) and the single-core split-K kernel types the reference declares
// (op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36).
//
// Why registers instead of the LDS ring: a decode call is bound by bytes in flight per CU and by fixed costs (launch, first
// round trip, combine launch: profiles/r03_decode_cold.txt).  A wave holds D k blocks of fragments in registers -- 8 waves x D x
// (16 TM + 16 TN) rows x 128 B = 200-300 KB in flight per CU against the 72 KB of a three-stage LDS ring -- and the 512 KB
// register file is the one buffer a CU has that is larger than its LDS.
//
// Work split.  Workgroup w of G owns n-tiles (16 columns each) [w nt / G, (w + 1) nt / G) -- at most TNMAX of them, balanced to
// within one, so 18432 columns on 256 CUs are 4 or 5 n-tiles each instead of 288 fixed tiles in two rounds -- and ALL rows
// (M <= 16 TM).  Wave s owns k blocks [s kbps, (s + 1) kbps), kbps = ceil(kb_n / 8): the same slices, the same per-slice
// arithmetic (one v_mfma_scale_f32_16x16x128_f8f6f4 per scale block, fp32 promotion) and the same combine order (s ascending) as
// the two-launch split-K with splitkFactor 8 (gemm_fp8_kernel.hpp + splitk_reduce_bf16_kernel): the outputs are bit-identical
// to it, which is what tests/test_wsk_gpu.py asserts.
// A is re-read by every workgroup (from L2: it is M x K bytes in all).
//
// Three builds in this file.  gemm_fp8_wsk_kernel (below): fragments global -> registers, M <= 64 -- never the fastest kernel of its
// cold sweep (a fragment load takes half a cache line per request and its loads are ones the compiler counts), kept selectable
// (tiling.stages = 1).  gemm_fp8_wskd_kernel: the operands staged through per-wave LDS-DMA rings, M <= 32 -- what kernelSerial 6 runs
// and what the tuned table and the selector pick for decode rows (profiles/r04_sweep_wskd/).  gemm_fp8_wskc_kernel: the same with a
// ring that runs on across the passes of a workgroup that walks more n-tiles than one pass holds (wide matrices).
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

template <int TM, int TNMAX, int D, bool KTAIL>
__global__ void __launch_bounds__(512) gemm_fp8_wsk_kernel(const GemmParams p)
{
    constexpr int WAVES = 8, BNW = TNMAX * 16, BM = TM * 16;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];   // [WAVES][BM][BNW] fp32
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    constexpr uint32_t kOutOfRange = 0x80000000u;

    const int M = p.m;
    const int nt_total = (p.n + 15) >> 4, G = gridDim.x;
    const int nt0 = (int)(((int64_t)blockIdx.x * nt_total) / G), nt1 = (int)(((int64_t)(blockIdx.x + 1) * nt_total) / G);
    const int cnt = nt1 - nt0;                      // <= TNMAX (host)
    if (cnt <= 0) return;
    const int n0 = nt0 * 16;
    const int kbps = (p.kb_n + WAVES - 1) / WAVES;
    const int c0 = wave * kbps, c1 = min(p.kb_n, c0 + kbps);   // this wave's k blocks (empty: c0 >= c1)
    const int s_eff = (p.kb_n + kbps - 1) / kbps;              // waves that own at least one k block

    auto clamp31 = [](int64_t v) { return (int)(v > 0x7FFFFFFFll ? 0x7FFFFFFFll : (v < 0 ? 0 : v)); };
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.a, 0, clamp31((int64_t)M * p.lda), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(p.b + (int64_t)n0 * p.ldb), 0, clamp31((int64_t)(p.n - n0) * p.ldb), 0x00020000);
    const __amdgpu_buffer_rsrc_t sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.sfa, 0, clamp31((int64_t)M * p.sfa_ld * 4), 0x00020000);
    // the workgroup's columns lie in at most two 128-wide scale blocks (TNMAX * 16 <= 128)
    static_assert(TNMAX * 16 <= 128, "two sfb blocks per workgroup at most");
    const int nb0 = n0 >> 7, nb1 = min(nb0 + 1, p.nb_n - 1);
    const float *sfb0_p = p.sfb + (int64_t)nb0 * p.kb_n, *sfb1_p = p.sfb + (int64_t)nb1 * p.kb_n;
    const __amdgpu_buffer_rsrc_t sfb_rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.sfb, 0, clamp31((int64_t)p.nb_n * p.kb_n * 4), 0x00020000);
    const int sfb0_off = nb0 * p.kb_n * 4, sfb1_off = nb1 * p.kb_n * 4;
    (void)sfb0_p; (void)sfb1_p;

    // fragment rows: lane (li, kg) holds bytes [16 kg, +16) and [64 + 16 kg, +16) of row li of its tile's k block
    uint32_t a_voff[TM], b_voff[TNMAX], sfa_voff[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int row = mt * 16 + li;
        a_voff[mt] = row < M ? (uint32_t)row * (uint32_t)p.lda + 16 * kg : kOutOfRange;
        sfa_voff[mt] = row < M ? (uint32_t)row * (uint32_t)p.sfa_ld * 4u : kOutOfRange;
    }
#pragma unroll
    for (int t = 0; t < TNMAX; ++t) {
        const int row = t * 16 + li;   // relative to n0
        b_voff[t] = (t < cnt && n0 + row < p.n) ? (uint32_t)row * (uint32_t)p.ldb + 16 * kg : kOutOfRange;
    }
    // which scale block an n-tile lies in (wave-uniform per t)
    bool second_block[TNMAX];
#pragma unroll
    for (int t = 0; t < TNMAX; ++t) second_block[t] = ((n0 + t * 16) >> 7) != nb0;

    struct Stage {
        v8i a[TM], b[TNMAX];
        float sfa[TM], sfb0, sfb1;
    };
    Stage st[D];
#ifndef DGA_WSK_B_AUX
#define DGA_WSK_B_AUX 0   // default policy: a fragment load takes HALF of each 128-byte line, the second half comes from the L1 line the first one filled -- with nt (2) the line is fetched twice: 8x18432x7168 cold 29.9 -> 34.0 us
#endif
    auto load16 = [&](const __amdgpu_buffer_rsrc_t &rs, uint32_t vo, int kb, int half, auto auxc) -> v4i {
        // a lane whose chunk lies beyond K (KTAIL) or whose k block lies beyond the wave's slice reads nothing (zeros)
        const int k0 = kb * 128 + half * 64;
        bool ok = kb < c1;
        if constexpr (KTAIL) ok = ok && (k0 + 16 * kg < p.k);
        const uint32_t v = ok ? vo : kOutOfRange;
        return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)v, k0, decltype(auxc)::value));
    };
    auto load_stage = [&](Stage &s, int kb) {
        const int kbc = min(kb, p.kb_n - 1);
#pragma unroll
        for (int t = 0; t < TNMAX; ++t) {
            const v4i lo = load16(b_rsrc, b_voff[t], kb, 0, std::integral_constant<int, DGA_WSK_B_AUX>{}), hi = load16(b_rsrc, b_voff[t], kb, 1, std::integral_constant<int, DGA_WSK_B_AUX>{});
            s.b[t] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const v4i lo = load16(a_rsrc, a_voff[mt], kb, 0, std::integral_constant<int, 0>{}), hi = load16(a_rsrc, a_voff[mt], kb, 1, std::integral_constant<int, 0>{});
            s.a[mt] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            s.sfa[mt] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sfa_rsrc, (int)sfa_voff[mt], kbc * 4, 0));
        }
        s.sfb0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sfb_rsrc, 0, sfb0_off + kbc * 4, 0));
        s.sfb1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sfb_rsrc, 0, sfb1_off + kbc * 4, 0));
    };

    v4f acc[TM][TNMAX];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TNMAX; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int d = 0; d < D; ++d) load_stage(st[d], c0 + d);
    for (int kb = c0; kb < c1; kb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            if (kb + d < c1) {   // (wave-uniform)
#pragma unroll
                for (int t = 0; t < TNMAX; ++t) {
                    if (t < cnt) {
                        const float sb = second_block[t] ? st[d].sfb1 : st[d].sfb0;
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt) {
                            const v4f pr = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(st[d].b[t], st[d].a[mt], v4f{0.f, 0.f, 0.f, 0.f},
                                                                                           0, 0, 0, 0, 0, 0);
                            const float s = st[d].sfa[mt] * sb;   // two-level scale: sfa[m, kb] * sfb[n / 128, kb]
                            acc[mt][t].x = __builtin_fmaf(pr.x, s, acc[mt][t].x);
                            acc[mt][t].y = __builtin_fmaf(pr.y, s, acc[mt][t].y);
                            acc[mt][t].z = __builtin_fmaf(pr.z, s, acc[mt][t].z);
                            acc[mt][t].w = __builtin_fmaf(pr.w, s, acc[mt][t].w);
                        }
                    }
                }
            }
            load_stage(st[d], kb + d + D);   // (beyond the slice: every lane out of range, nothing is fetched)
        }
    }

    // ---- the eight partial tiles meet in LDS: slab[wave][m][n] fp32; lane (li, kg) owns row m = 16 mt + li, columns 16 t + 4 kg + [0, 4)
    float *slab = (float *)smem;
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int t = 0; t < TNMAX; ++t)
            *(v4f *)(slab + ((size_t)wave * BM + mt * 16 + li) * BNW + t * 16 + 4 * kg) = acc[mt][t];
    __syncthreads();
    // s ascending = k ascending, as splitk_reduce_bf16_kernel sums its slabs
    const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)p.out) & 7) == 0);
    for (int g = tid; g < BM * (BNW / 4); g += 512) {
        const int m = g / (BNW / 4), nl = (g % (BNW / 4)) * 4;
        if (m >= M || nl >= cnt * 16) continue;
        v4f v = *(const v4f *)(slab + (size_t)m * BNW + nl);
        for (int s = 1; s < s_eff; ++s) v += *(const v4f *)(slab + ((size_t)s * BM + m) * BNW + nl);
        const v2bf h0 = __builtin_convertvector(v2f{v.x, v.y}, v2bf), h1 = __builtin_convertvector(v2f{v.z, v.w}, v2bf);
        uint16_t *dst = p.out + (int64_t)m * p.ldc + n0 + nl;
        if (vec_ok && n0 + nl + 4 <= p.n) {
            typedef int v2i __attribute__((ext_vector_type(2)));
            *(v2i *)dst = v2i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1)};
        } else {
            const uint32_t w0 = __builtin_bit_cast(uint32_t, h0), w1 = __builtin_bit_cast(uint32_t, h1);
            const uint16_t e[4] = {(uint16_t)(w0 & 0xFFFFu), (uint16_t)(w0 >> 16), (uint16_t)(w1 & 0xFFFFu), (uint16_t)(w1 >> 16)};
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (n0 + nl + q < p.n) dst[q] = e[q];
        }
    }
}

// ---- the same split with the operands staged through LDS-DMA -------------------------------------------------------------------
// What sank the register build on cold weights (profiles/r04_decode_cold.txt): a fragment load takes HALF of a 128-byte line per
// row and request, and its loads are ones the compiler counts.  Here every wave owns a private ring of D stages in LDS -- a stage =
// the 16 TM rows of A and 16 TN rows of B of one k block, 128 bytes each, plus their scales -- filled by LDS-DMA (whole lines: a
// wave-instruction moves 8 rows x 128 bytes) and read back as fragments by ordinary ds_read_b128.  Nothing is shared between the
// waves while they stream (no barrier); every vector-memory instruction of the loop is inline-asm DMA, so the only vmcnt waits are
// the hand-placed ones (one per k block: the oldest stage has landed, the D - 1 younger ones stay in flight).  The LDS image is the
// tile kernels': chunk c of row r at chunk position c ^ ((r >> 1) & 7), conflict-free for the fragment reads.
// A workgroup owns the n-tiles [w nt / G, (w + 1) nt / G) as above and walks them TN at a time; each pass re-streams the A rows
// (from L2).  Same slices, same per-slice arithmetic, same combine order: bit-identical to the register build and to the
// two-launch split-K with splitkFactor WAVES.  LDS: WAVES x D x (2 (TM + TN) KB + 256 B); the partial tiles meet in the rings.
// WAVES = 8 (M <= 32) or 4 (M <= 64: half as many K slices, twice the ring per wave -- what 64 rows of A per stage need).
// MATH = 1: the bf16-exact policy's arithmetic (dispatchPolicyTag 7) on the same rings -- the fragments are up-converted in
// registers (v_cvt_scalef32_pk_bf16_fp8, exact) and a scale block is four chained v_mfma_f32_16x16x32_bf16 with the k placement of
// the tile kernel's MATH = 1 loop (gemm_fp8_kernel.hpp): bit-identical to that policy's two-launch split-K.  The stream pays for
// neither the conversions nor the second matrix rate.
template <int TM, int TN, int D, bool KTAIL, int WAVES = 8, int MATH = 0>
__global__ void __launch_bounds__(WAVES * 64) gemm_fp8_wskd_kernel(const GemmParams p)
{
    constexpr int BM = TM * 16, BNW = TN * 16, ROWS = BM + BNW, NT = WAVES * 64;
    constexpr int SCI = (BM + 2 + 63) / 64;                  // scale DMA instructions per stage (BM sfa rows + 2 sfb words, 64 per instruction)
    constexpr int L = ROWS / 8 + SCI;                        // DMA instructions per stage: 8 rows each, + the scales
    constexpr int STAGE = ROWS * 128 + SCI * 256, RING = D * STAGE;
    static_assert(WAVES * RING <= 160 * 1024 && BM * BNW * 4 <= RING, "LDS of one CU; a wave's partial tile fits its own ring");
    static_assert((D - 1) * L < 64, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    constexpr uint32_t kOutOfRange = 0x80000000u;

    const int M = p.m;
    const int nt_total = (p.n + 15) >> 4, G = gridDim.x;
    const int nt0 = (int)(((int64_t)blockIdx.x * nt_total) / G), nt1 = (int)(((int64_t)(blockIdx.x + 1) * nt_total) / G);
    if (nt1 <= nt0) return;
    const int kbps = (p.kb_n + WAVES - 1) / WAVES;
    const int c0 = wave * kbps, c1 = min(p.kb_n, c0 + kbps);   // this wave's k blocks (empty: c0 >= c1)
    const int s_eff = (p.kb_n + kbps - 1) / kbps;              // waves that own at least one k block

    uint8_t *ring = smem + wave * RING;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(lptr_t)smem + wave * RING;
    // DMA sources.  Instruction j of a stage fills image rows 8 j .. 8 j + 7: lane -> row 8 j + (lane >> 3), source chunk
    // (lane & 7) ^ x(row).  Rows [0, BM) are A rows, rows [BM, ROWS) the pass's B rows.
    const int d_row = lane >> 3;
    const v4i a_rsrc = make_rsrc(p.a, (int64_t)M * p.lda);
    uint32_t a_voff[BM / 8];
    int col[2];   // (the swizzle looks at the row inside its 16-row tile: instruction j's column depends on j & 1 only)
#pragma unroll
    for (int j = 0; j < 2; ++j) col[j] = ((lane & 7) ^ swz_a((8 * j + d_row) & 15)) * 16;
#pragma unroll
    for (int j = 0; j < BM / 8; ++j) {
        const int row = 8 * j + d_row;
        a_voff[j] = row < M ? (uint32_t)row * (uint32_t)p.lda + col[j & 1] : kOutOfRange;
    }
    // fragment reads: lane (li, kg) takes chunks kg and 4 + kg of row li of its tile
    const int f_off0 = li * 128 + ((kg ^ swz_a(li)) * 16), f_off1 = li * 128 + (((4 + kg) ^ swz_a(li)) * 16);
    // scales of a stage: one 4-byte DMA per lane and instruction -- slots [0, BM): sfa of row `slot`; slots BM, BM + 1: the two
    // sfb blocks of the pass; the other slots re-fetch the first sfb word

    for (int ntc = nt0; ntc < nt1; ntc += TN) {
        const int cnt = min(TN, nt1 - ntc), n0 = ntc * 16;
        const v4i b_rsrc = make_rsrc(p.b + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
        uint32_t b_voff[BNW / 8];
#pragma unroll
        for (int j = 0; j < BNW / 8; ++j) {
            const int row = 8 * j + d_row;   // relative to n0
            b_voff[j] = (row < cnt * 16 && n0 + row < p.n) ? (uint32_t)row * (uint32_t)p.ldb + col[j & 1] : kOutOfRange;
        }
        const int nb0 = n0 >> 7, nb1 = min(nb0 + 1, p.nb_n - 1);
        const float *sc_src[SCI];
#pragma unroll
        for (int i = 0; i < SCI; ++i) {
            const int slot = i * 64 + lane;
            sc_src[i] = slot < BM ? p.sfa + (int64_t)min(slot, M - 1) * p.sfa_ld : p.sfb + (int64_t)(slot == BM + 1 ? nb1 : nb0) * p.kb_n;
        }
        bool second_block[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) second_block[t] = ((n0 + t * 16) >> 7) != nb0;

        // one stage: k block kb of the wave's slice into ring stage `stg`; beyond the slice every lane is out of range (zeros
        // land, nothing is fetched) so that the number of instructions in flight stays what the waits assume
        auto issue_stage = [&](int stg, int kb) {
            const uint32_t base = ring_lds + stg * STAGE;
            const bool live = kb < c1;
            // (starting the workgroups' slices at different k blocks -- so that they do not all ask for the same A lines at once -- changes
            //  nothing: 16 x 7168 x 18432 31.8 against 32.0 us; A costs its bytes, not a hot spot)
            const int k0 = kb * 128;
            bool ok2[2] = {live, live};
            if constexpr (KTAIL) {
                ok2[0] = live && (k0 + col[0] < p.k);
                ok2[1] = live && (k0 + col[1] < p.k);
            }
#pragma unroll
            for (int j = 0; j < ROWS / 8; ++j) {
                uint32_t vo = j < BM / 8 ? a_voff[j < BM / 8 ? j : 0] : b_voff[j >= BM / 8 ? j - BM / 8 : 0];
                vo = ok2[j & 1] ? vo : kOutOfRange;
                dma16(vo, j < BM / 8 ? a_rsrc : b_rsrc, (uint32_t)k0, base + j * 1024);
            }
#pragma unroll
            for (int i = 0; i < SCI; ++i) dma4(sc_src[i] + min(kb, p.kb_n - 1), base + ROWS * 128 + i * 256);
        };

        v4f acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int d = 0; d < D; ++d) issue_stage(d, c0 + d);
        for (int kb = c0; kb < c1; kb += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (kb + d < c1) {   // (wave-uniform; once false it stays false, so no later wait depends on the skipped refill)
                    wait_vmcnt<(D - 1) * L>();   // stage d has landed; the D - 1 younger stages stay in flight
                    const uint8_t *st = ring + d * STAGE;
                    v8i af[TM], bf[TN];
                    float sfa_r[TM];
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        const v4i lo = *(const v4i *)(st + mt * 2048 + f_off0), hi = *(const v4i *)(st + mt * 2048 + f_off1);
                        af[mt] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                        sfa_r[mt] = *(const float *)(st + ROWS * 128 + (mt * 16 + li) * 4);
                    }
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        const v4i lo = *(const v4i *)(st + (TM + t) * 2048 + f_off0), hi = *(const v4i *)(st + (TM + t) * 2048 + f_off1);
                        bf[t] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
                    const float sfb0 = *(const float *)(st + ROWS * 128 + BM * 4), sfb1 = *(const float *)(st + ROWS * 128 + BM * 4 + 4);
                    // the stage is in registers: refill it (the reads must have returned before the DMA may overwrite it)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    issue_stage(d, kb + d + D);
                    // MATH = 1: conversion c (0..15) of a fragment: dword c >> 1 of its 32 bytes, half c & 1 -> dword c & 3 of MFMA c >> 2
                    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
                    v4i afx[MATH ? TM : 1][4], bfx[MATH ? TN : 1][4];
                    if constexpr (MATH == 1) {
                        auto convert = [](const v8i &raw, v4i (&dst)[4]) {
#pragma unroll
                            for (int c = 0; c < 16; ++c) {
                                const int w = raw[c >> 1];
                                dst[c >> 2][c & 3] = (c & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                                                             : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
                            }
                        };
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt) convert(af[mt], afx[mt]);
#pragma unroll
                        for (int t = 0; t < TN; ++t) convert(bf[t], bfx[t]);
                    }
#pragma unroll
                    for (int t = 0; t < TN; ++t) {
                        if (t < cnt) {
                            const float sb = second_block[t] ? sfb1 : sfb0;
#pragma unroll
                            for (int mt = 0; mt < TM; ++mt) {
                                v4f pr;
                                if constexpr (MATH == 1) {
                                    pr = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                                    for (int q = 0; q < 4; ++q)
                                        pr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, bfx[t][q]),
                                                                                     __builtin_bit_cast(v8bf, afx[mt][q]), pr, 0, 0, 0);
                                } else
                                pr = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[t], af[mt], v4f{0.f, 0.f, 0.f, 0.f},
                                                                                               0, 0, 0, 0, 0, 0);
                                const float s = sfa_r[mt] * sb;   // two-level scale: sfa[m, kb] * sfb[n / 128, kb]
                                acc[mt][t].x = __builtin_fmaf(pr.x, s, acc[mt][t].x);
                                acc[mt][t].y = __builtin_fmaf(pr.y, s, acc[mt][t].y);
                                acc[mt][t].z = __builtin_fmaf(pr.z, s, acc[mt][t].z);
                                acc[mt][t].w = __builtin_fmaf(pr.w, s, acc[mt][t].w);
                            }
                        }
                    }
                }
            }
        }
        wait_vmcnt<0>();   // the refills beyond the slice (zeros) have landed: the ring is this wave's to reuse

        // ---- the eight partial tiles meet in LDS, each in its wave's own ring: slab[m][n] fp32; lane (li, kg) owns row
        //      m = 16 mt + li, columns 16 t + 4 kg + [0, 4)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int t = 0; t < TN; ++t) *(v4f *)((float *)ring + (mt * 16 + li) * BNW + t * 16 + 4 * kg) = acc[mt][t];
        __syncthreads();
        // s ascending = k ascending, as splitk_reduce_bf16_kernel sums its slabs
        const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)p.out) & 7) == 0);
        for (int g = tid; g < BM * (BNW / 4); g += NT) {
            const int m = g / (BNW / 4), nl = (g % (BNW / 4)) * 4;
            if (m >= M || nl >= cnt * 16) continue;
            v4f v = *(const v4f *)((const float *)smem + m * BNW + nl);
            for (int s = 1; s < s_eff; ++s) v += *(const v4f *)((const float *)(smem + s * RING) + m * BNW + nl);
            const v2bf h0 = __builtin_convertvector(v2f{v.x, v.y}, v2bf), h1 = __builtin_convertvector(v2f{v.z, v.w}, v2bf);
            uint16_t *dst = p.out + (int64_t)m * p.ldc + n0 + nl;
            if (vec_ok && n0 + nl + 4 <= p.n) {
                typedef int v2i __attribute__((ext_vector_type(2)));
                *(v2i *)dst = v2i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1)};
            } else {
                const uint32_t w0 = __builtin_bit_cast(uint32_t, h0), w1 = __builtin_bit_cast(uint32_t, h1);
                const uint16_t e[4] = {(uint16_t)(w0 & 0xFFFFu), (uint16_t)(w0 >> 16), (uint16_t)(w1 & 0xFFFFu), (uint16_t)(w1 >> 16)};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n0 + nl + q < p.n) dst[q] = e[q];
            }
        }
        if (ntc + TN < nt1) {   // the next pass refills the rings: every wave has read the partial tiles, and this wave's stores
            __syncthreads();    // (counted in vmcnt like the DMA) are out of the way of the next pass's waits
            wait_vmcnt<0>();
        }
    }
}

// ---- the LDS-DMA build for workgroups that walk MORE than TN n-tiles: one continuous ring ------------------------------------------
// gemm_fp8_wskd_kernel drains its rings at the end of every pass (the partial tiles meet IN the rings) and starts the next pass with
// nothing in flight: three passes on an 18432-row matrix are three first round trips, which is why it only ties the tile kernels
// there.  Here the partial tiles have a slab of their own behind the rings and the refill never stops: a wave's stages are the
// flattened (pass, k block) sequence, the stage a wave has just read is refilled with the block D positions further on -- the same
// pass or the next -- and a pass boundary costs the two barriers around the combine, not a round trip.  Everything else (image,
// hand-counted vmcnt, slices, arithmetic, combine order) is gemm_fp8_wskd_kernel's: the same bits.  The global stores of a combine
// are counted in vmcnt like the DMA; they only make the next waits stricter (vector-memory operations retire in order).
template <int TM, int TN, int D, bool KTAIL, int MATH = 0>
__global__ void __launch_bounds__(512) gemm_fp8_wskc_kernel(const GemmParams p)
{
    constexpr int WAVES = 8, BM = TM * 16, BNW = TN * 16, ROWS = BM + BNW, NT = WAVES * 64;
    constexpr int SCI = (BM + 2 + 63) / 64, L = ROWS / 8 + SCI;
    constexpr int STAGE = ROWS * 128 + SCI * 256, RING = D * STAGE, SLAB = BM * BNW * 4;
    static_assert(WAVES * (RING + SLAB) <= 160 * 1024, "LDS of one CU");
    static_assert((D - 1) * L < 64, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    constexpr uint32_t kOutOfRange = 0x80000000u;

    const int M = p.m;
    const int nt_total = (p.n + 15) >> 4, G = gridDim.x;
    const int nt0 = (int)(((int64_t)blockIdx.x * nt_total) / G), nt1 = (int)(((int64_t)(blockIdx.x + 1) * nt_total) / G);
    if (nt1 <= nt0) return;
    const int npass = (nt1 - nt0 + TN - 1) / TN;
    const int kbps = (p.kb_n + WAVES - 1) / WAVES;
    const int c0 = wave * kbps, c1 = min(p.kb_n, c0 + kbps);
    const int len = max(0, c1 - c0);                            // k blocks of this wave's slice
    const int s_eff = (p.kb_n + kbps - 1) / kbps;

    uint8_t *ring = smem + wave * RING;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(lptr_t)smem + wave * RING;
    float *slab = (float *)(smem + WAVES * RING);               // [wave][m][n] fp32
    const int d_row = lane >> 3;
    const v4i a_rsrc = make_rsrc(p.a, (int64_t)M * p.lda);
    uint32_t a_voff[BM / 8];
    int col[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) col[j] = ((lane & 7) ^ swz_a((8 * j + d_row) & 15)) * 16;
#pragma unroll
    for (int j = 0; j < BM / 8; ++j) {
        const int row = 8 * j + d_row;
        a_voff[j] = row < M ? (uint32_t)row * (uint32_t)p.lda + col[j & 1] : kOutOfRange;
    }
    const int f_off0 = li * 128 + ((kg ^ swz_a(li)) * 16), f_off1 = li * 128 + (((4 + kg) ^ swz_a(li)) * 16);

    // ---- the refill cursor: pass ip, k block ik of the slice, ring stage istg; D positions ahead of the multiplication
    int ip = 0, ik = 0, istg = 0;
    v4i b_rsrc = a_rsrc;
    uint32_t b_voff[BNW / 8];
    const float *sc_src[SCI];
    auto set_issue_pass = [&](int pass) {
        const int ntc = nt0 + pass * TN, cnt = min(TN, nt1 - ntc), n0 = ntc * 16;
        b_rsrc = make_rsrc(p.b + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
#pragma unroll
        for (int j = 0; j < BNW / 8; ++j) {
            const int row = 8 * j + d_row;
            b_voff[j] = (row < cnt * 16 && n0 + row < p.n) ? (uint32_t)row * (uint32_t)p.ldb + col[j & 1] : kOutOfRange;
        }
        const int nb0 = n0 >> 7, nb1 = min(nb0 + 1, p.nb_n - 1);
#pragma unroll
        for (int i = 0; i < SCI; ++i) {
            const int slot = i * 64 + lane;
            sc_src[i] = slot < BM ? p.sfa + (int64_t)min(slot, M - 1) * p.sfa_ld : p.sfb + (int64_t)(slot == BM + 1 ? nb1 : nb0) * p.kb_n;
        }
    };
    // the next stage of the sequence (past its end: every lane out of range -- zeros land, nothing is fetched -- so that the
    // number of instructions in flight stays what the waits assume)
    auto issue_next = [&]() {
        const uint32_t base = ring_lds + istg * STAGE;
        const bool live = ip < npass;
        const int kb = c0 + ik, k0 = kb * 128;
        bool ok2[2] = {live, live};
        if constexpr (KTAIL) {
            ok2[0] = live && (k0 + col[0] < p.k);
            ok2[1] = live && (k0 + col[1] < p.k);
        }
#pragma unroll
        for (int j = 0; j < ROWS / 8; ++j) {
            uint32_t vo = j < BM / 8 ? a_voff[j < BM / 8 ? j : 0] : b_voff[j >= BM / 8 ? j - BM / 8 : 0];
            vo = ok2[j & 1] ? vo : kOutOfRange;
            dma16(vo, j < BM / 8 ? a_rsrc : b_rsrc, (uint32_t)k0, base + j * 1024);
        }
#pragma unroll
        for (int i = 0; i < SCI; ++i) dma4(sc_src[i] + min(kb, p.kb_n - 1), base + ROWS * 128 + i * 256);
        istg = istg + 1 == D ? 0 : istg + 1;
        if (live && ++ik == len) {
            ik = 0;
            if (++ip < npass) set_issue_pass(ip);
        }
    };
    set_issue_pass(0);
    if (len > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue_next();
    }

    int cstg = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int ntc = nt0 + pass * TN, cnt = min(TN, nt1 - ntc), n0 = ntc * 16;
        const int nb0 = n0 >> 7;
        bool second_block[TN];
#pragma unroll
        for (int t = 0; t < TN; ++t) second_block[t] = ((n0 + t * 16) >> 7) != nb0;
        v4f acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < len; ++i) {
            wait_vmcnt<(D - 1) * L>();   // the oldest stage has landed; the D - 1 younger ones stay in flight
            const uint8_t *st = ring + cstg * STAGE;
            v8i af[TM], bf[TN];
            float sfa_r[TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const v4i lo = *(const v4i *)(st + mt * 2048 + f_off0), hi = *(const v4i *)(st + mt * 2048 + f_off1);
                af[mt] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                sfa_r[mt] = *(const float *)(st + ROWS * 128 + (mt * 16 + li) * 4);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const v4i lo = *(const v4i *)(st + (TM + t) * 2048 + f_off0), hi = *(const v4i *)(st + (TM + t) * 2048 + f_off1);
                bf[t] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
            const float sfb0 = *(const float *)(st + ROWS * 128 + BM * 4), sfb1 = *(const float *)(st + ROWS * 128 + BM * 4 + 4);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is in registers: refill it
            issue_next();
            cstg = cstg + 1 == D ? 0 : cstg + 1;
            typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
            v4i afx[MATH ? TM : 1][4], bfx[MATH ? TN : 1][4];
            if constexpr (MATH == 1) {
                auto convert = [](const v8i &raw, v4i (&dst)[4]) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const int w = raw[c >> 1];
                        dst[c >> 2][c & 3] = (c & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                                                     : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
                    }
                };
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) convert(af[mt], afx[mt]);
#pragma unroll
                for (int t = 0; t < TN; ++t) convert(bf[t], bfx[t]);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (t < cnt) {
                    const float sb = second_block[t] ? sfb1 : sfb0;
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) {
                        v4f pr;
                        if constexpr (MATH == 1) {
                            pr = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                            for (int q = 0; q < 4; ++q)
                                pr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, bfx[t][q]),
                                                                             __builtin_bit_cast(v8bf, afx[mt][q]), pr, 0, 0, 0);
                        } else
                        pr = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[t], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                        const float s = sfa_r[mt] * sb;
                        acc[mt][t].x = __builtin_fmaf(pr.x, s, acc[mt][t].x);
                        acc[mt][t].y = __builtin_fmaf(pr.y, s, acc[mt][t].y);
                        acc[mt][t].z = __builtin_fmaf(pr.z, s, acc[mt][t].z);
                        acc[mt][t].w = __builtin_fmaf(pr.w, s, acc[mt][t].w);
                    }
                }
            }
        }
        // ---- the pass's partial tiles meet in the slab behind the rings (the rings keep streaming the next pass)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int t = 0; t < TN; ++t) *(v4f *)(slab + (size_t)wave * BM * BNW + (mt * 16 + li) * BNW + t * 16 + 4 * kg) = acc[mt][t];
        __syncthreads();
        const bool vec_ok = ((p.ldc & 3) == 0) && ((((uintptr_t)p.out) & 7) == 0);
        for (int g = tid; g < BM * (BNW / 4); g += NT) {
            const int m = g / (BNW / 4), nl = (g % (BNW / 4)) * 4;
            if (m >= M || nl >= cnt * 16) continue;
            v4f v = *(const v4f *)(slab + m * BNW + nl);
            for (int s = 1; s < s_eff; ++s) v += *(const v4f *)(slab + (size_t)s * BM * BNW + m * BNW + nl);
            const v2bf h0 = __builtin_convertvector(v2f{v.x, v.y}, v2bf), h1 = __builtin_convertvector(v2f{v.z, v.w}, v2bf);
            uint16_t *dst = p.out + (int64_t)m * p.ldc + n0 + nl;
            if (vec_ok && n0 + nl + 4 <= p.n) {
                typedef int v2i __attribute__((ext_vector_type(2)));
                *(v2i *)dst = v2i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1)};
            } else {
                const uint32_t w0 = __builtin_bit_cast(uint32_t, h0), w1 = __builtin_bit_cast(uint32_t, h1);
                const uint16_t e[4] = {(uint16_t)(w0 & 0xFFFFu), (uint16_t)(w0 >> 16), (uint16_t)(w1 & 0xFFFFu), (uint16_t)(w1 >> 16)};
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n0 + nl + q < p.n) dst[q] = e[q];
            }
        }
        if (pass + 1 < npass) __syncthreads();   // everyone has read the slab before the next pass's partial tiles go there
    }
    wait_vmcnt<0>();   // the refills past the sequence (zeros) have landed
}

}  // namespace dga
