// One-launch split-K for decode rows under the bf16-exact policy (dispatchPolicyTag 7, kernelSerial 6, build DGA_BUILD_BX_DECODE): dense
// problems of a few 64-row tiles whose weight matrix is too narrow to give every CU a tile (64 x 4096 x 7168: 32 tiles of 64 x 128).
//
// What the two-launch split-K pays there (profiles/r06_decode_stamps.txt: 15.6 us for 5 us of weight stream): a lone wave per SIMD that
// exposes every conversion -> MFMA -> promotion dependency (1988 ticks per k block for 512 of matrix pipe), seven k blocks per
// workgroup, 8.4 MB of fp32 slabs, a kernel boundary behind them and a combine launch.  Here
//   * a workgroup is TWO k groups of four waves on the same 64 x 128 tile: group h of workgroup s takes k slice 2 s + h of 2 S, on its
//     own three-stage LDS-DMA ring -- two waves per SIMD that hide each other's latencies with 64 x 32 wave tiles (the 8-wave tile
//     build has 32 x 32 ones: a conversion more per MFMA), and half the k blocks per wave;
//   * the two groups' accumulators meet in LDS, the S workgroups' in the caller's workspace (write-through rows + a flag that holds the
//     launch's epoch and that its reader puts back to 0: nothing to zero in front of a launch, captured or not); the workgroup with the first slices adds them in k order and stores
//     the bf16 rows: half the partial bytes of a split by 2 S, no second launch.  All workgroups are resident at once (launcher).
// The loop is the masked grouped kernel's (gemm_fp8_bf16x_grouped_kernel.hpp: self-contained k blocks, two of them in flight, the
// refill on the MFMA gaps), walked m-tile-major so that the A conversions spread evenly over a block's gaps.
//
// Arithmetic: the policy's (four chained v_mfma_f32_16x16x32_bf16 per scale block on exactly converted operands, one fp32 promotion
// per block, k ascending inside a slice); the slices are summed as ((g0 + g1) + (g2 + g3)) + ... -- a fixed order: the result does
// not depend on timing, and differs from the sequential builds' by fp32 rounding of the partial sums only (tests/test_bf16x_dsk_gpu.py).
//
// (Measured, not kept: every workgroup of a tile adding its 1 / S of the tile over all S partial tiles -- the reads of the partials spread
//  over S CUs, flags per workgroup with a generation word -- instead of one adding workgroup: +0.7 ... +1.2 us at S = 2..6, level at 7 and
//  8 (same box, two passes each: the adder's own partial write and S - 1 flags to watch per workgroup cost more than its read burst); the
//  refill in a burst behind the barrier: +2 ... +8 %.)
//
// Reference counterparts: the split-K kernel types the reference declares (op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36), the
// fused reduce of its Stream-K kernel (op_kernel/kernel/padding_streamk_matmul_kernel.h:92-107) and the Small handler that serves
// these shapes there (op_host/op_tiling/select_kernel.cpp:270-291).
#pragma once
#include "gemm_fp8_kernel.hpp"
#include "gemm_fp8_streamk_kernel.hpp"   // StreamKArgs

namespace dga {

struct DskCfg {
    static constexpr int BM = 64, BN = 128, GT = 256, NT = 512, TN = 2;
    static constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, SC_BYTES = GT * 4;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES + SC_BYTES, RING_BYTES = 3 * STAGE_BYTES, LDS_BYTES = 2 * RING_BYTES;
    // the A-image build: two stages + the bf16 image of one k block's A tile (m-tile, chain step, lane: 16 bytes each) per k group
    static constexpr int IMG_BYTES = BM * 128 * 2, IMG_GROUP_BYTES = 2 * STAGE_BYTES + IMG_BYTES, IMG_LDS_BYTES = 2 * IMG_GROUP_BYTES;
    static constexpr int A_ITERS = 2, B_ITERS = 4, NL = A_ITERS + B_ITERS + 1;
    static constexpr int SLOT_FLOATS = BM * BN;     // one workgroup's partial tile
    static constexpr int MAX_S = 8;
};

// IMG: the A tile of a k block is converted ONCE per k group -- wave w its m-tile w, into a bf16 image in LDS behind a second barrier --
// instead of once per wave: 16 + 32 conversions per wave and k block instead of 64 + 32 (the loop is bound by the vector port: two waves
// per SIMD take twice one wave's time), and the raw stage is dead at that barrier, so two stages carry the refill as far ahead as three.
template <bool KTAIL, bool IMG = true>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_fp8_bf16x_dsk_kernel(const GemmParams p, const StreamKArgs sk)
{
    typedef DskCfg C;
    constexpr int BM = C::BM, BN = C::BN, GT = C::GT, TN = C::TN, NL = C::NL, LAGT = 2, RING = 4;
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63, gtid = tid & (GT - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = wave >> 2, gw = wave & 3;       // k group, wave of the group (its 32 columns)
    const int li = lane & 15, kg = lane >> 4;
    const int KB = p.kb_n, S = p.splitk;          // S >= 1; KB >= 4 S (host): every slice has two k blocks at least

    // ---- workgroup -> (m-tile, n-tile, split): the m-tiles of one (n-tile, split) -- they read the same weight bytes -- lie on one XCD
    int v = blockIdx.x;
    if (p.xcd_remap) {
        const int nwg = gridDim.x, xcd = v & 7, q = nwg >> 3, r = nwg & 7;
        v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    }
    const int tm = v % p.tiles_m, rest = v / p.tiles_m, s = rest % S, tn = rest / S;
    const int tile = tn * p.tiles_m + tm;
    const int M = p.m, m0 = tm * BM, n0 = tn * BN;
    // slice i of 2 S: k blocks [cut(i), cut(i + 1)); the adding workgroup's two slices weigh wa, the others wo (host: every slice has two
    // blocks at least) -- its partners' partials are then on their way while it still multiplies
    const int wa = p.tail_sub > 0 ? (p.tail_sub & 0xFF) : 1, wo = p.tail_sub > 0 ? (p.tail_sub >> 8) : 1;
    const int W = 2 * wa + (2 * S - 2) * wo;
    auto cut = [&](int i) { return (int)(((int64_t)KB * (i <= 2 ? i * wa : 2 * wa + (i - 2) * wo)) / W); };
    const int i0 = 2 * s + h, io = 2 * s + (1 - h);
    const int kb0 = cut(i0), kb1 = cut(i0 + 1);
    const int nblk = kb1 - kb0;
    const int nblk_o = cut(io + 1) - cut(io);
    const int nmax = max(nblk, nblk_o);           // barriers are the workgroup's: both groups walk the longer slice's count

    // ---- LDS-DMA sources of this group's ring (gemm_fp8_kernel.hpp: chunk c = it * GT + gtid lands at byte 16 c of the image)
    uint8_t *const ring = smem + h * (IMG ? C::IMG_GROUP_BYTES : C::RING_BYTES);
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)ring;
    constexpr uint32_t kOutOfRange = 0x80000000u;
    const int a_col = ((gtid & 7) ^ swz_a(gtid >> 3)) * 16;
    const int b_col = ((gtid & 7) ^ swz_b(gtid >> 3)) * 16;
    uint32_t a_voff[C::A_ITERS], b_voff[C::B_ITERS];
#pragma unroll
    for (int it = 0; it < C::A_ITERS; ++it) {
        const int row = (it * GT + gtid) >> 3;
        a_voff[it] = row < M - m0 ? (uint32_t)row * (uint32_t)p.lda + a_col : kOutOfRange;    // rows beyond M: zero-filled, not fetched
    }
#pragma unroll
    for (int it = 0; it < C::B_ITERS; ++it) {
        const int row = (it * GT + gtid) >> 3;
        b_voff[it] = (uint32_t)min(row, p.n - 1 - n0) * (uint32_t)p.ldb + b_col;
    }
    const v4i a_rsrc = make_rsrc(p.a + (int64_t)m0 * p.lda, (int64_t)(M - m0) * p.lda);
    const v4i b_rsrc = make_rsrc(p.b + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
    // scale slots: [0, BM) the tile's sfa rows, BM its sfb block (the rest of the piece re-reads it)
    const float *const sc_src = gtid < BM ? p.sfa + (int64_t)min(m0 + gtid, M - 1) * p.sfa_ld
                                          : p.sfb + (int64_t)min(n0 / 128, p.nb_n - 1) * p.kb_n;
    // piece idx of k block kb into `stage`; past the slice: every lane out of range (zeros land, nothing is fetched)
    auto refill = [&](int idx, int stage, int kb) {
#ifdef DGA_DSK_KNOBS
        if (p.tail_begin & 8) return;       // (diagnostic: nothing is fetched)
#endif
        const uint32_t sa = lds0 + stage * C::STAGE_BYTES + gw * 1024;
        const int k0 = kb * 128;
        const bool live = kb < kb1;
        if (idx < C::A_ITERS) {
            uint32_t voff = live ? a_voff[idx] : kOutOfRange;
            if constexpr (KTAIL) voff = (k0 + a_col < p.k) ? voff : kOutOfRange;
            dma16(voff, a_rsrc, (uint32_t)k0, sa + idx * GT * 16);
        } else if (idx < C::A_ITERS + C::B_ITERS) {
            const int it = idx - C::A_ITERS;
            uint32_t voff = live ? b_voff[it] : kOutOfRange;
            if constexpr (KTAIL) voff = (k0 + b_col < p.k) ? voff : kOutOfRange;
            dma16(voff, b_rsrc, (uint32_t)k0, sa + C::A_BYTES + it * GT * 16);
        } else {
            dma4(sc_src + (live ? kb : kb1 - 1), lds0 + stage * C::STAGE_BYTES + C::A_BYTES + C::B_BYTES + gw * 256);
        }
    };

    // ---- per-lane fragment read offsets (bytes inside a stage): gemm_fp8_kernel.hpp
    const int a_off0 = li * 128 + ((kg ^ swz_a(li)) * 16);
    const int a_off1 = li * 128 + (((kg + 4) ^ swz_a(li)) * 16);
    const int b_row = gw * (BN / 4) + 8 * (li >> 2) + (li & 3);
    const int b_off0 = C::A_BYTES + b_row * 128 + ((kg ^ swz_b(b_row)) * 16);
    const int b_off1 = C::A_BYTES + b_row * 128 + (((kg + 4) ^ swz_b(b_row)) * 16);
    const int sa_off = C::A_BYTES + C::B_BYTES + li * 4;
    const int sb_off = C::A_BYTES + C::B_BYTES + BM * 4;

    v4f acc[4][TN];
    v4f part[RING];
    v4i afx[2][4], bfx[TN][4];      // bf16 fragments: [q] = the 8 bf16 of MFMA q of the chain; A of the m-tile at hand and the next one
    v4i braw[TN][2], araw[2][2];    // raw e4m3 bytes: [0] = bytes [16 kg, +16), [1] = bytes [64 + 16 kg, +16)
    float s_cur[4], s_old[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    auto convert = [](const v4i (&raw)[2], v4i (&dst)[4], int c) {
        const int w = raw[(c >> 1) >> 2][(c >> 1) & 3];
        dst[c >> 2][c & 3] = (c & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                                     : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
    };

    // ---- prologue: the slice's first THREE blocks on their way (every stage of the ring: a slice is a handful of blocks, and the
    //      first round trip from a cold HBM is the longest) -- block 0 then issues no refill, its predecessor's stage being block 2's
#pragma unroll
    for (int d = 0; d < (IMG ? 2 : 3); ++d)       // (the image build has two stages)
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) refill(idx, d, kb0 + d);

    // the slice with L m-tiles of 16 rows (2: at most 32 rows; 4)
    auto run = [&](auto Lc) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr int TILES = L * TN, GAPS = 4 * TILES, SP = L;
        static_assert(TILES % RING == 0 && 1 + (NL - 1) * SP < GAPS, "the ring of partial tiles; the refill fits a block");
#pragma unroll
        for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < L; ++i) s_old[i] = 0.f;     // the first LAGT tiles "promote the previous block": part (= 0) * 0
        int cur = 0;
        for (int it = 0; it < nmax; ++it) {
            if (it == 0) wait_vmcnt<2 * NL>();   // this wave's pieces of block `it` have landed (the younger blocks' may be in flight)
            else wait_vmcnt<NL>();
            __builtin_amdgcn_s_barrier();    // everyone's have; and everyone has left block it - 1, whose stage is refilled now
            asm volatile("" ::: "memory");
#ifdef DGA_DSK_KNOBS
            if ((p.tail_begin & 4) && it < nblk) {      // (diagnostic: the refill alone, nothing multiplied)
                if (it > 0) for (int idx = 0; idx < NL; ++idx) refill(idx, cur >= 1 ? cur - 1 : 2, kb0 + it + 2);
            } else
#endif
            if (it < nblk) {
                const int fill = cur >= 1 ? cur - 1 : 2, kbf = kb0 + it + 2;
                const uint8_t *sc = ring + cur * C::STAGE_BYTES;
#ifdef DGA_DSK_KNOBS
                if ((p.tail_begin & 32) && it > 0)       // (experiment: the refill in a burst behind the barrier)
                    for (int idx = 0; idx < NL; ++idx) refill(idx, fill, kbf);
#endif
                // the block's first fragments out of its stage: both B tiles and A(0) converted in a burst
                araw[0][0] = *(const v4i *)(sc + a_off0);
                araw[0][1] = *(const v4i *)(sc + a_off1);
                braw[0][0] = *(const v4i *)(sc + b_off0);
                braw[0][1] = *(const v4i *)(sc + b_off1);
                braw[1][0] = *(const v4i *)(sc + b_off0 + 512);
                braw[1][1] = *(const v4i *)(sc + b_off1 + 512);
                araw[1][0] = *(const v4i *)(sc + a_off0 + 2048);
                araw[1][1] = *(const v4i *)(sc + a_off1 + 2048);
                const float sfb0 = *(const float *)(sc + sb_off);
                float sa[L];
#pragma unroll
                for (int i = 0; i < L; ++i) sa[i] = *(const float *)(sc + sa_off + i * 64);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 16; ++c) convert(braw[0], bfx[0], c);
#pragma unroll
                for (int c = 0; c < 16; ++c) convert(araw[0], afx[0], c);
                if constexpr (L > 2) {
                    araw[0][0] = *(const v4i *)(sc + a_off0 + 2 * 2048);
                    araw[0][1] = *(const v4i *)(sc + a_off1 + 2 * 2048);
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) convert(braw[1], bfx[1], c);
#pragma unroll
                for (int i = 0; i < L; ++i) s_cur[i] = sa[i] * sfb0;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GAPS; ++u) {
                    const int t = u >> 2, q = u & 3, mt = t / TN, nt = t % TN, g8 = u % (4 * TN);
                    part[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(v8bf, bfx[nt][q]), __builtin_bit_cast(v8bf, afx[mt & 1][q]),
                        q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t % RING], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#ifdef DGA_DSK_KNOBS
                    if (!(p.tail_begin & 32))
#endif
                    if (u >= 1 && (u - 1) % SP == 0 && (u - 1) / SP < NL && it > 0) refill((u - 1) / SP, fill, kbf);
                    // A(mt + 1) is converted behind the MFMAs of m-tile mt (two conversions a gap); its raw bytes were read an m-tile earlier
                    if (mt + 1 < L) {
#pragma unroll
                        for (int c = 2 * g8; c < 2 * g8 + 2; ++c) convert(araw[(mt + 1) & 1], afx[(mt + 1) & 1], c);
                        if (g8 == 4 * TN - 1 && mt + 3 < L) {
                            araw[(mt + 1) & 1][0] = *(const v4i *)(sc + a_off0 + (mt + 3) * 2048);
                            araw[(mt + 1) & 1][1] = *(const v4i *)(sc + a_off1 + (mt + 3) * 2048);
                        }
                    }
                    {   // the promotion of tile t - LAGT (the previous block's last tiles during this block's first ones)
                        const int j = t >= LAGT ? t - LAGT : TILES + t - LAGT, jm = j / TN, jn = j % TN;
                        const float sv = t >= LAGT ? s_cur[jm] : s_old[jm];
                        acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], sv, acc[jm][jn][q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < L; ++i) s_old[i] = s_cur[i];
            }
            cur = cur == 2 ? 0 : cur + 1;
        }
        // the last LAGT tiles of the last block
#pragma unroll
        for (int t = 0; t < LAGT; ++t) {
            const int j = TILES + t - LAGT, jm = j / TN, jn = j % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], s_old[jm], acc[jm][jn][q]);
        }
    };
    // the same slice on the A-image build: per k block  [barrier: the block has landed]  convert: A m-tile `gw` -> image, both B tiles ->
    // registers  [barrier: the image is whole; the raw stage is dead]  multiply, the refill of block it + 2 -- into the stage just
    // consumed -- on the MFMA gaps.  (The second k group one interval behind the first -- one wave of a SIMD converting while the other
    // multiplies -- measured 11 % SLOWER: a lone wave does not fill the matrix pipe, and the pair waits for the longer interval.)
    auto run_img = [&](auto Lc) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr int TILES = L * TN, GAPS = 4 * TILES, SP = L;
        static_assert(TILES % RING == 0 && 1 + (NL - 1) * SP < GAPS, "the ring of partial tiles; the refill fits a block");
        uint8_t *const img = ring + 2 * C::STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < L; ++i) s_old[i] = 0.f;
        int cur = 0;
        for (int it = 0; it < nmax; ++it) {
            wait_vmcnt<NL>();                // this wave's pieces of block `it` have landed (the younger block's may be in flight)
            __builtin_amdgcn_s_barrier();    // everyone's have; and everyone has left block it - 1: its image may be overwritten
            asm volatile("" ::: "memory");
            const bool live = it < nblk;
            const uint8_t *sc = ring + cur * C::STAGE_BYTES;
            if (live) {
                if (gw < L) {
                    araw[0][0] = *(const v4i *)(sc + a_off0 + gw * 2048);
                    araw[0][1] = *(const v4i *)(sc + a_off1 + gw * 2048);
                }
                braw[0][0] = *(const v4i *)(sc + b_off0);
                braw[0][1] = *(const v4i *)(sc + b_off1);
                braw[1][0] = *(const v4i *)(sc + b_off0 + 512);
                braw[1][1] = *(const v4i *)(sc + b_off1 + 512);
                const float sfb0 = *(const float *)(sc + sb_off);
                float sa[L];
#pragma unroll
                for (int i = 0; i < L; ++i) sa[i] = *(const float *)(sc + sa_off + i * 64);
                if (gw < L) {
#pragma unroll
                    for (int c = 0; c < 16; ++c) convert(araw[0], afx[0], c);
#pragma unroll
                    for (int q = 0; q < 4; ++q) *(v4i *)(img + ((gw * 4 + q) * 64 + lane) * 16) = afx[0][q];
                }
#pragma unroll
                for (int c = 0; c < 16; ++c) convert(braw[0], bfx[0], c);
#pragma unroll
                for (int c = 0; c < 16; ++c) convert(braw[1], bfx[1], c);
#pragma unroll
                for (int i = 0; i < L; ++i) s_cur[i] = sa[i] * sfb0;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();    // the image is whole -- and nobody reads the raw stage any more
            asm volatile("" ::: "memory");
            if (live) {
                const int kbf = kb0 + it + 2;
#pragma unroll
                for (int q = 0; q < 4; ++q) afx[0][q] = *(const v4i *)(img + (q * 64 + lane) * 16);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < GAPS; ++u) {
                    const int t = u >> 2, q = u & 3, mt = t / TN, nt = t % TN;
                    part[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(v8bf, bfx[nt][q]), __builtin_bit_cast(v8bf, afx[mt & 1][q]),
                        q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t % RING], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (u >= 1 && (u - 1) % SP == 0 && (u - 1) / SP < NL) refill((u - 1) / SP, cur, kbf);
                    // A(mt + 1) out of the image behind the first MFMA of m-tile mt (its registers were A(mt - 1)'s)
                    if (nt == 0 && q == 0 && mt + 1 < L) {
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) afx[(mt + 1) & 1][qq] = *(const v4i *)(img + (((mt + 1) * 4 + qq) * 64 + lane) * 16);
                    }
                    {   // the promotion of tile t - LAGT (the previous block's last tiles during this block's first ones)
                        const int j = t >= LAGT ? t - LAGT : TILES + t - LAGT, jm = j / TN, jn = j % TN;
                        const float sv = t >= LAGT ? s_cur[jm] : s_old[jm];
                        acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], sv, acc[jm][jn][q]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int i = 0; i < L; ++i) s_old[i] = s_cur[i];
            }
            cur ^= 1;
        }
#pragma unroll
        for (int t = 0; t < LAGT; ++t) {
            const int j = TILES + t - LAGT, jm = j / TN, jn = j % TN;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], s_old[jm], acc[jm][jn][q]);
        }
    };
    if constexpr (IMG) {
        if (M - m0 <= 32) run_img(std::integral_constant<int, 2>{});
        else run_img(std::integral_constant<int, 4>{});
    } else {
        if (M - m0 <= 32) run(std::integral_constant<int, 2>{});
        else run(std::integral_constant<int, 4>{});
    }

    // ---- the two groups' accumulators meet in LDS (lane-linear: tile ti of wave gw at float4 (gw * 8 + ti) * 64 + lane)
    wait_vmcnt<0>();                 // the refills past the slice (zeros) have landed: the ring is dead from here on
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    v4f *const X = reinterpret_cast<v4f *>(smem);
    if (h == 1) {
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) X[(gw * 8 + ti) * 64 + lane] = acc[ti >> 1][ti & 1];
    }
    __syncthreads();
    if (h == 0) {
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) {
            const v4f o = X[(gw * 8 + ti) * 64 + lane];
            v4f &a = acc[ti >> 1][ti & 1];
            a.x += o.x; a.y += o.y; a.z += o.z; a.w += o.w;
        }
    }
    uint16_t *const Cout = p.out;
    const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)Cout) & 15) == 0);
    auto store_row = [&](int m, int n, const v4f &lo, const v4f &hi) {
        if (m >= M) return;
        uint16_t *crow = Cout + (int64_t)m * p.ldc;
        const v2bf h0 = __builtin_convertvector(v2f{lo.x, lo.y}, v2bf);
        const v2bf h1 = __builtin_convertvector(v2f{lo.z, lo.w}, v2bf);
        const v2bf h2 = __builtin_convertvector(v2f{hi.x, hi.y}, v2bf);
        const v2bf h3 = __builtin_convertvector(v2f{hi.z, hi.w}, v2bf);
        const v4i pk = v4i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1), __builtin_bit_cast(int, h2), __builtin_bit_cast(int, h3)};
        if (vec_ok && n + 8 <= p.n) {
            *(v4i *)(crow + n) = pk;
        } else {
            const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (n + q < p.n) crow[n + q] = e[q];
        }
    };
    if (S == 1) {
        if (h == 0) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) store_row(m0 + 16 * mt + li, n0 + gw * (BN / 4) + 8 * kg, acc[mt][0], acc[mt][1]);
        }
        return;
    }
#ifdef DGA_DSK_KNOBS
    const int knob = p.tail_begin;
    if ((knob & 1) && s > 0) return;      // (diagnostic: the partials are neither written nor waited for)
#endif
    if (s > 0) {
        // ---- a partial: this workgroup's sum goes to its slot (write-through), then the flag behind every wave's drained stores
        const int slot = tile * (S - 1) + (s - 1);
        if (h == 0) {
            float *dst = sk.partials + (int64_t)slot * C::SLOT_FLOATS + (gw * 8 * 64 + lane) * 4;
#pragma unroll
            for (int ti = 0; ti < 8; ++ti) {
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(acc[ti >> 1][ti & 1]) : "memory");
                dst += 64 * 4;
                asm volatile("" : "+v"(dst));
            }
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid == 0) __hip_atomic_store(sk.flags + slot, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    // ---- the adding workgroup (the first slices): its own sum back to LDS, then all 512 threads add the S - 1 partials in k order --
    //      thread (wave w, lane) owns the two 16-byte output pieces (m-tile 2 (w >> 2) + {0, 1}, wave column w & 3, lane)
    if (h == 0) {
#pragma unroll
        for (int ti = 0; ti < 8; ++ti) X[(gw * 8 + ti) * 64 + lane] = acc[ti >> 1][ti & 1];
    }
    // one thread watches the flags, one partial at a time: the reads of a partial are on their way while the next one's flag is still
    // awaited (the partners do not finish together: after the last flag only the last partial's 32 KB are left to come, not all of them).
    // Behind the last barrier the flags are put back to 0 -- a launch replayed from a graph repeats its epoch, and finds the flags as an
    // ordinary launch does: not holding it (no memset node in front of the kernel).
    const int e0 = (gw * 8 + 4 * h) * 64 + lane;      // float4 index of (m-tile 2 h, n-tile 0); + 64 per following tile
    v4f w[C::MAX_S - 1][4];
    __syncthreads();                                  // (the adding workgroup's own sum is in X)
#pragma unroll
    for (int j = 0; j < C::MAX_S - 1; ++j) {
        if (j < S - 1) {
#ifdef DGA_DSK_KNOBS
            if (!(knob & 3))
#endif
            if (tid == 0)
                while (__hip_atomic_load(sk.flags + tile * (S - 1) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(2);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            // (the watching wave reads its own share of the partials behind the last flag: a load of its own in flight would stand in
            //  front of its next look at a flag)
            if (wave != 0) {
                const float *src = sk.partials + (int64_t)(tile * (S - 1) + j) * C::SLOT_FLOATS + e0 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(w[j][e]) : "v"(src) : "memory");
                    src += 64 * 4;
                    asm volatile("" : "+v"(src));
                }
            }
        }
    }
    if (wave == 0) {
#pragma unroll
        for (int j = 0; j < C::MAX_S - 1; ++j) {
            if (j < S - 1) {
                const float *src = sk.partials + (int64_t)(tile * (S - 1) + j) * C::SLOT_FLOATS + e0 * 4;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(w[j][e]) : "v"(src) : "memory");
                    src += 64 * 4;
                    asm volatile("" : "+v"(src));
                }
            }
        }
    }
    if (tid == 0)
        for (int j = 0; j < S - 1; ++j) __hip_atomic_store(sk.flags + tile * (S - 1) + j, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    v4f sum[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) sum[e] = X[e0 + 64 * e];
    wait_vmcnt<0>();
#pragma unroll
    for (int j = 0; j < C::MAX_S - 1; ++j) {
        if (j < S - 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                asm volatile("" : "+v"(w[j][e]));     // (the loads' results are valid from here on)
                sum[e].x += w[j][e].x; sum[e].y += w[j][e].y; sum[e].z += w[j][e].z; sum[e].w += w[j][e].w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) store_row(m0 + 16 * (2 * h + i) + li, n0 + gw * (BN / 4) + 8 * kg, sum[2 * i], sum[2 * i + 1]);
}

}  // namespace dga
