// bf16-exact policy (dispatchPolicyTag 7), image build: 128 x 256 tile, ONE wave per SIMD, both operands shared through a bf16
// LDS image that is converted ONCE per workgroup.
//
// Same arithmetic as the MATH = 1 loop of gemm_fp8_kernel.hpp (bit-identical outputs: the same four chained
// v_mfma_f32_16x16x32_bf16 per 128-wide scale block with the same k placement, the same fp32 promotion), i.e. the CDNA4
// counterpart of the reference's device K-loop (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:123-369)
// held to the fp32-accumulate golden of /root/reference/deep_gemm_ascend/framework/tests/test.py:19-64.  What changes is who
// converts.  In the MATH = 1 loop every wave converts the e4m3 fragments it multiplies -- A fragments by the 4 waves of a tile
// row, B by the 2 of a column: 1024 v_cvt_scalef32_pk_bf16_fp8 per k block per CU for 384 unique ones, two per MFMA beside the
// promotion FMA, and the loop is bound by vector issue.  Here
//   * each wave fetches a quarter of the tile's e4m3 bytes to registers (buffer_load_dwordx4, one k block ahead of the
//     conversion, two ahead of the MFMAs), converts them once (exact) and ds_write_b128s a swizzled bf16 image:
//     96 conversions per wave per k block for 128 MFMAs = 0.75 per MFMA;
//   * the 4 waves (2 x 2, wave tile 64 x 128) read bf16 fragments with ds_read_b128 straight into MFMA operand registers
//     (AGPRs: no vector instruction touches them).  The B-matrix fragments of a k block (8 n-tiles) are HELD for the whole
//     block, the A-matrix fragments (4 m-tiles) stream through two buffers one tile row ahead.
// LDS: the held operand's image is dead once every wave has read it (first quarter of the block), so it is single-buffered
// (64 KB); the streamed A image is double-buffered (2 x 32 KB): 128 KB.  Two barriers per k block:
//   Y (after tile row 0): every wave has the block's B fragments in registers -> the B image may be overwritten;
//   X (7/8 into the block): the next block's images are complete -> its first fragments are read under the block's last MFMAs.
// Image layout (both operands): image row r = 256 B = 16 units of 16 B; the unit that lane (li = r & 15, kg) reads for MFMA q
// of the chain sits at unit position rotl1_4((4 q | kg) ^ li): a ds_read_b128's four 16-lane groups and a ds_write_b128's eight
// 8-lane groups each touch distinct banks.  B image rows are permuted so that the 16 rows of an n-tile fragment are consecutive
// (source row 32 j + 8 a + 4 h + b of a wave's 128 -> image row 16 (2 j + h) + 4 a + b; see the orientation note of
// gemm_fp8_kernel.hpp).
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

// WAVES = 4: 2 x 2 waves, wave tile 64 x 128, one wave per SIMD (B-matrix fragments of 8 n-tiles held, 160 AGPRs).
// WAVES = 8: 2 x 4 waves, wave tile 64 x 64, two waves per SIMD (4 n-tiles held; every register of the 256 a wave may own is
//            spoken for) -- the partner wave's matrix and vector work covers this wave's LDS / memory / wait issue slots.
// The static schedule of one k block lives here too: G MFMA "gaps" per wave, u = 4 * tile + q, tile = TN * mt + nt; it is shared
// by the kernel body and by the hand-placed waits.  LDS operations riding on gap u, in issue order:
// [image store][B read][A read][next-block read].
template <int WAVES>
struct BxImageCfg {
    static_assert(WAVES == 4 || WAVES == 8, "2 x 2 or 2 x 4 waves");
    static constexpr int kBM = 128, kBN = 256, NT = WAVES * 64;
    static constexpr int kWM = 2, kWN = WAVES / 2;
    static constexpr int TM = 4, TN = kBN / kWN / 16;     // wave tile 64 x (128 | 64)
    static constexpr int A_IMG = 128 * 256, B_IMG = 256 * 256;
    static constexpr int LDS_BYTES = 2 * A_IMG + B_IMG;   // A stage 0 | A stage 1 | B
    static constexpr int PIECE_ROWS = NT / 8;             // 1 piece = one dwordx4 per thread = PIECE_ROWS rows x 128 B
    static constexpr int A_PIECES = kBM / PIECE_ROWS, B_PIECES = kBN / PIECE_ROWS, PIECES = A_PIECES + B_PIECES;
    static constexpr int TILES = TM * TN, G = 4 * TILES;
    // conversions: one per gap; the A pieces (their stage of the A ring is free from the block's start) from gap CA0, the B pieces
    // behind barrier Y (gap YGAP: tile row 0 is done, every wave holds the block's B fragments); barrier X (gap XGAP) in front
    // of the last three tiles, whose gaps read the next block's first fragments
    static constexpr int CA0 = WAVES == 4 ? 4 : 2, YGAP = 4 * TN + (WAVES == 4 ? 4 : 0), CB0 = YGAP + (WAVES == 4 ? 4 : 2);
    static constexpr int XGAP = 4 * (TILES - 3);
    static_assert(CA0 + 8 * A_PIECES <= YGAP + 2 && CB0 + 8 * B_PIECES <= XGAP, "the image is complete in front of X");

    static constexpr bool gap_converts(int u) { return (u >= CA0 && u < CA0 + 8 * A_PIECES) || (u >= CB0 && u < CB0 + 8 * B_PIECES); }
    static constexpr int gap_ci(int u) { return u < CB0 ? (u - CA0) & 7 : (u - CB0) & 7; }
    static constexpr int gap_piece(int u) { return u < CB0 ? (u - CA0) >> 3 : A_PIECES + ((u - CB0) >> 3); }
    static constexpr bool gap_stores(int u) { return gap_converts(u) && (gap_ci(u) & 3) == 3; }
    static constexpr bool gap_reads_b(int u) { return (u >> 2) < TN - 2; }                                      // Bh[tile + 2][q]
    static constexpr bool gap_reads_a(int u) { return ((u >> 2) % TN) == TN / 2 && (u >> 2) / TN < TM - 1; }    // As[(mt + 1) & 1][q]
    static constexpr bool gap_reads_next(int u) { return (u >> 2) >= TILES - 3; }   // Bh[0], Bh[1], As[0] of the next block
    static constexpr int gap_ops(int u) { return gap_stores(u) + gap_reads_b(u) + gap_reads_a(u) + gap_reads_next(u); }
    static constexpr int ops_before(int u) { int n = 0; for (int v = 0; v < u; ++v) n += gap_ops(v); return n; }
    // index, in the block's sequence of LDS operations, of the read riding on gap v (kind 0: B, 1: A, 2: next-block)
    static constexpr int read_index(int v, int kind)
    {
        return ops_before(v) + gap_stores(v) + (kind >= 1 ? gap_reads_b(v) : 0) + (kind >= 2 ? gap_reads_a(v) : 0);
    }
    // lgkmcnt value in front of MFMA(u) that guarantees both its fragments have landed (LDS operations complete in order);
    // >= 16: no wait needed (an operand's first use in the block is the only one that can find it in flight)
    static constexpr int wait_for_gap(int u)
    {
        const int t = u >> 2, q = u & 3, mt = t / TN, nt = t % TN;
        const int issued = ops_before(u), per_block = ops_before(G);
        int w = 99;
        if (mt == 0) {   // first use of Bh[nt][q]
            const int left = nt >= 2 ? issued - (read_index(4 * (nt - 2) + q, 0) + 1)
                                     : issued + per_block - (read_index(4 * (TILES - 3 + nt) + q, 2) + 1);
            w = left < w ? left : w;
        }
        if (nt == 0) {   // first use of As[mt & 1][q]
            const int left = mt >= 1 ? issued - (read_index(4 * ((mt - 1) * TN + TN / 2) + q, 1) + 1)
                                     : issued + per_block - (read_index(4 * (TILES - 1) + q, 2) + 1);
            w = left < w ? left : w;
        }
        return w;
    }
};

namespace bximg {
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
}  // namespace bximg

__device__ __forceinline__ int rotl1_4(int x) { return ((x << 1) | (x >> 3)) & 15; }

template <class Cfg, bool KTAIL, bool CLK = false>
__global__ void __launch_bounds__(Cfg::NT) __attribute__((amdgpu_waves_per_eu(Cfg::NT / 256, Cfg::NT / 256)))
gemm_fp8_bf16x_image_kernel(const GemmParams p)
{
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, TM = Cfg::TM, TN = Cfg::TN, NT = Cfg::NT, TILES = Cfg::TILES;
    constexpr int WROWS = BM / Cfg::kWM, WCOLS = BN / Cfg::kWN;   // wave tile
    LoopClock<CLK> loop_clock;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    typedef unsigned v4u __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / Cfg::kWN, wn = wave % Cfg::kWN;
    const int li = lane & 15, kg = lane >> 4;

    // ---- tile id: XCD-aware remap, then the grouped raster (as gemm_fp8_blockscaled_nt_kernel; dense, masked grouped, split-K)
    const int nwg = gridDim.x, bid = blockIdx.x;
    int tile;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = p.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
    }
    const int tiles_per_group = p.tiles_m * p.tiles_n;
    const int split = p.splitk > 1 ? tile / tiles_per_group : 0;
    const int g = p.splitk > 1 ? 0 : tile / tiles_per_group;
    const int t_in = tile - (p.splitk > 1 ? split : g) * tiles_per_group;
    int tm, tn;
    {
        const int gm = p.raster_group;
        const int per = gm * p.tiles_n;
        const int band = t_in / per;
        const int first = band * gm;
        const int rows = min(p.tiles_m - first, gm);
        const int loc = t_in - band * per;
        tm = first + loc % rows;
        tn = loc / rows;
    }
    const int M = p.masked_m ? min(p.masked_m[g], p.m) : p.m;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;
    const int kb_begin = p.splitk > 1 ? split * p.kb_per_split : 0;
    const int kb_end = p.splitk > 1 ? min(p.kb_n, kb_begin + p.kb_per_split) : p.kb_n;
    if (kb_begin >= kb_end) return;

    const uint8_t *A = p.a + (int64_t)g * p.a_gs;
    const uint8_t *B = p.b + (int64_t)g * p.b_gs;
    const float *SFA = p.sfa + (int64_t)g * p.sfa_gs;
    const float *SFB = p.sfb + (int64_t)g * p.sfb_gs;
    uint16_t *C = p.out + (int64_t)g * p.c_gs;

    // ---- image writers: thread -> (row wrow of a piece, 16-byte chunk wc of the row's 128-byte k block)
    constexpr uint32_t kOutOfRange = 0x80000000u;
    constexpr int PR = Cfg::PIECE_ROWS;
    const int wrow = tid >> 3, wc = tid & 7;
    // Per-thread byte offset inside a piece; the piece's first row (PR * pc rows further) is a scalar added per load, and the
    // descriptors end at the tile's last valid row (A: row M - 1, B: row n - 1): a lane whose row lies beyond gets zeros from
    // the range check (such rows / columns are never stored).  KTAIL adds the per-lane beyond-K test.
    uint32_t a_voff = (uint32_t)wrow * (uint32_t)p.lda + 16 * wc, b_voff = (uint32_t)wrow * (uint32_t)p.ldb + 16 * wc;
    const uint32_t a_step = (uint32_t)PR * (uint32_t)p.lda, b_step = (uint32_t)PR * (uint32_t)p.ldb;
    auto clamp31 = [](int64_t v) { return (int)(v > 0x7FFFFFFFll ? 0x7FFFFFFFll : (v < 0 ? 0 : v)); };
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(A + (int64_t)m0 * p.lda), 0, clamp31((int64_t)(M - m0) * p.lda), 0x00020000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(B + (int64_t)n0 * p.ldb), 0, clamp31((int64_t)(p.n - n0) * p.ldb), 0x00020000);
    auto load_piece = [&](int pc, int kb) -> v4i {
        const int k0 = kb * 128;
        uint32_t vo = pc < Cfg::A_PIECES ? a_voff + pc * a_step : b_voff + (pc - Cfg::A_PIECES) * b_step;
        if constexpr (KTAIL) vo = (k0 + 16 * wc < p.k) ? vo : kOutOfRange;
#ifdef DGA_BXI_NOLOAD   // diagnostic: nothing is fetched
        v4i z = v4i{(int)vo, k0, 0, 0};
        asm volatile("" : "+v"(z));
        return z;
#else
        const v4u r = __builtin_amdgcn_raw_buffer_load_b128(pc < Cfg::A_PIECES ? a_rsrc : b_rsrc, (int)vo, k0, 0);
        return __builtin_bit_cast(v4i, r);
#endif
    };
    // unit position of (chunk c, half j) in an image row whose low four row bits are l
    auto unit = [](int c, int j, int l) { return rotl1_4((8 * (c >> 2) + 4 * j + (c & 3)) ^ l); };
    // B image row of source row wrow of a piece: inside every 32 rows, [a:2][h][b:2] -> [h][a:2][b:2]
    const int brow = (wrow & ~31) + ((wrow >> 2) & 1) * 16 + ((wrow >> 3) & 3) * 4 + (wrow & 3);
    // LDS byte offsets of this thread's two 16-byte stores per piece (piece pc adds pc * PR * 256); the A stage toggles per k block
    uint32_t a_wr0 = Cfg::A_IMG + wrow * 256 + unit(wc, 0, wrow & 15) * 16;
    uint32_t a_wr1 = Cfg::A_IMG + wrow * 256 + unit(wc, 1, wrow & 15) * 16;
    const uint32_t b_wr0 = 2 * Cfg::A_IMG + brow * 256 + unit(wc, 0, brow & 15) * 16;
    const uint32_t b_wr1 = 2 * Cfg::A_IMG + brow * 256 + unit(wc, 1, brow & 15) * 16;
    // conversion ci (0..7) of a piece's 16 raw bytes: dword ci >> 1, half ci & 1 -> dword ci of the 32 bf16 bytes
    v4i cv[2];
    auto convert = [&](const v4i &raw, int ci) {
        const int w = raw[ci >> 1];
#ifdef DGA_BXI_NOCVT    // diagnostic: no conversion (the raw bytes are stored as they are)
        cv[ci >> 2][ci & 3] = w;
        return;
#endif
        cv[ci >> 2][ci & 3] = (ci & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                                       : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
    };
    // LDS traffic is inline asm on purpose: (a) a fragment read must land in AGPRs ("=a": left to the allocator the fragments
    // take VGPRs first and the accumulators spill through v_accvgpr moves, 118 extra vector instructions per k block), and
    // (b) the waits are placed by hand from the static schedule (Cfg::wait_for_gap) -- one in-order lgkmcnt counter covers
    // the reads and the image stores.
    auto store_half = [&](auto pcc, auto jc) __attribute__((always_inline)) {
        constexpr int pc = decltype(pcc)::value, j = decltype(jc)::value;
        constexpr int piece = pc < Cfg::A_PIECES ? pc : pc - Cfg::A_PIECES;
        const uint32_t base = pc < Cfg::A_PIECES ? (j ? a_wr1 : a_wr0) : (j ? b_wr1 : b_wr0);
        const v4i data = cv[j];   // (a local: clang rejects asm operands that name captures of a generic lambda)
#ifdef DGA_BXI_NOSTORE   // diagnostic (results are garbage): the image is never written
        asm volatile("" ::"v"(base), "v"(data));
#else
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(base), "v"(data), "n"(piece * PR * 256) : "memory");
#endif
    };
#ifdef DGA_BXI_NOREAD   // diagnostic: no fragment is read (whatever the registers hold is multiplied)
#define DGA_BX_READ_ASM(f, a, off) asm volatile("" : "=a"(f) : "v"(a))
#else
#define DGA_BX_READ_ASM(f, a, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(f) : "v"(a), "n"(off) : "memory")
#endif
#define DGA_BX_READ(dst, addr, off)                                                                              \
    do {                                                                                                         \
        v4i frag_;                                                                                               \
        const uint32_t addr_ = (addr);                                                                           \
        DGA_BX_READ_ASM(frag_, addr_, off);                                                                      \
        (dst) = frag_;                                                                                           \
    } while (0)

    // ---- fragment readers: lane (li, kg) reads unit (q, kg) of image row 16 * tile + li
    uint32_t a_rd[4], b_rd[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int u = rotl1_4((4 * q + kg) ^ li) * 16;
        a_rd[q] = (wm * WROWS + li) * 256 + u;                       // + mt * 4096; the stage toggles per k block
        b_rd[q] = 2 * Cfg::A_IMG + (wn * WCOLS + li) * 256 + u;     // + nt * 4096
    }

    // ---- scales: sfa of this lane's row of every m-tile, sfb of the wave's 128-wide n block; one block ahead in registers
    //      (buffer loads: rows at or beyond M read as zero scales; the sfb word is a scalar-addressed load)
    uint32_t sfa_voff = (uint32_t)li * (uint32_t)p.sfa_ld * 4u;
    const uint32_t sfa_step = 16u * (uint32_t)p.sfa_ld * 4u;
    const __amdgpu_buffer_rsrc_t sfa_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(SFA + (int64_t)(m0 + wm * WROWS) * p.sfa_ld), 0, clamp31((int64_t)(M - m0 - wm * WROWS) * p.sfa_ld * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t sfb_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(SFB + (int64_t)min((n0 + wn * WCOLS) / 128, p.nb_n - 1) * p.kb_n), 0, p.kb_n * 4, 0x00020000);
    auto load_sfa = [&](int mt, int kb) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sfa_rsrc, (int)(sfa_voff + mt * sfa_step), kb * 4, 0));
    };
    auto load_sfb = [&](int kb) { return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sfb_rsrc, 0, kb * 4, 0)); };

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    static_assert(TILES % 4 == 0 && TM == 4, "the part ring and the scale slots line up across k blocks");
    v4f part[4];   // ring: tile t's chain, tile t - 1 (finishing), tile t - 2 (being promoted)
#pragma unroll
    for (int i = 0; i < 4; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v4i Bh[TN][4], As[2][4], raw[Cfg::PIECES];
    float s_cur[TM], s_nxt[TM], s_old3 = 0.f, sfb_nxt = 0.f;

    auto barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS stores / reads are done
#ifndef DGA_BXI_NOBAR    // diagnostic: the waves run free
        __builtin_amdgcn_s_barrier();
#endif
        asm volatile("" ::: "memory");
    };

    // ---- prologue: block kb_begin fetched, converted and written; block kb_begin + 1 on its way; first fragments read
    const int kb_last = kb_end - 1;
#pragma unroll
    for (int pc = 0; pc < Cfg::PIECES; ++pc) raw[pc] = load_piece(pc, kb_begin);
    {
        const float sfb0 = load_sfb(kb_begin);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            s_cur[mt] = load_sfa(mt, kb_begin) * sfb0;
            s_nxt[mt] = 0.f;
        }
    }
    a_wr0 ^= Cfg::A_IMG; a_wr1 ^= Cfg::A_IMG;   // block kb_begin's A image goes to stage 0
    bximg::static_for<0, Cfg::PIECES>([&](auto pcc) __attribute__((always_inline)) {
        constexpr int pc = decltype(pcc)::value;
#pragma unroll
        for (int ci = 0; ci < 8; ++ci) convert(raw[pc], ci);
        store_half(pcc, std::integral_constant<int, 0>{});
        store_half(pcc, std::integral_constant<int, 1>{});
        asm volatile("s_nop 0" ::: "memory");   // (the stores' data registers are rewritten by the next piece's conversions)
        raw[pc] = load_piece(pc, min(kb_begin + 1, kb_last));
    });
    a_wr0 ^= Cfg::A_IMG; a_wr1 ^= Cfg::A_IMG;
    barrier();
    // (the order the loop's tail issues them in: the wait counts of the first block's first gaps assume it)
#pragma unroll
    for (int q = 0; q < 4; ++q) DGA_BX_READ(Bh[0][q], b_rd[q], 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) DGA_BX_READ(Bh[1][q], b_rd[q], 4096);
#pragma unroll
    for (int q = 0; q < 4; ++q) DGA_BX_READ(As[0][q], a_rd[q], 0);
    DGA_STAMP_DECL
    DGA_STAMP_CLOCK(6, 7);
    DGA_STAMP_START();
    loop_clock.tick();

    for (int kb = kb_begin; kb < kb_end; ++kb) {
        const int kb_scale = min(kb + 1, kb_last), kb_fetch = min(kb + 2, kb_last);
        // (opaque to the optimiser: the per-piece offsets are added where they are used -- hoisted out of the loop they would
        //  occupy more registers of a loop that has none to spare)
        asm volatile("" : "+v"(a_voff), "+v"(b_voff), "+v"(sfa_voff));
        bximg::static_for<0, Cfg::G>([&](auto uc) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            constexpr int t = u >> 2, q = u & 3, mt = t / TN, nt = t % TN;
            if constexpr (u == Cfg::YGAP) {   // Y: every wave holds this block's B fragments -> the B image may be overwritten
                DGA_STAMP(0);
                barrier();
                DGA_STAMP(1);
            }
            if constexpr (u == Cfg::XGAP) {   // X: the next block's images are complete
                DGA_STAMP(2);
                barrier();
                DGA_STAMP(3);
            }
            // the first MFMA that takes a fragment waits for its read.  (The fragment is not passed through the wait: an asm
            // that may write AGPRs in front of an MFMA costs a hazard s_nop each time; the sched_barrier keeps the order, and
            // scripts/check_bximg_waits.py checks it in the ISA.)
            if constexpr (Cfg::wait_for_gap(u) < 16) {
                asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(Cfg::wait_for_gap(u)) : "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            part[t & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                __builtin_bit_cast(v8bf, Bh[nt][q]), __builtin_bit_cast(v8bf, As[mt & 1][q]),
                q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // image of block kb + 1: one conversion per gap; a store after every fourth conversion; the piece's registers are
            // refilled with block kb + 2 as soon as its last conversion has read them
            if constexpr (Cfg::gap_converts(u)) {
                constexpr int pc = Cfg::gap_piece(u), ci = Cfg::gap_ci(u);
                convert(raw[pc], ci);
                if constexpr (ci == 3) store_half(std::integral_constant<int, pc>{}, std::integral_constant<int, 0>{});
                if constexpr (ci == 7) {
                    store_half(std::integral_constant<int, pc>{}, std::integral_constant<int, 1>{});
                    raw[pc] = load_piece(pc, kb_fetch);
                }
            }
            if constexpr (u == Cfg::CA0 + 8 * Cfg::A_PIECES) a_wr0 ^= Cfg::A_IMG;       // (behind the last A store)
            if constexpr (u == Cfg::CA0 + 8 * Cfg::A_PIECES + 1) a_wr1 ^= Cfg::A_IMG;
            // fragment reads: the rest of this block's B n-tiles two tiles ahead (row 0); the next A m-tile in the middle of
            // every tile row; behind X the next block's first fragments from the images just completed
            if constexpr (Cfg::gap_reads_b(u)) DGA_BX_READ(Bh[t + 2][q], b_rd[q], (t + 2) * 4096);
            if constexpr (Cfg::gap_reads_a(u)) DGA_BX_READ(As[(mt + 1) & 1][q], a_rd[q], (mt + 1) * 4096);
            if constexpr (t == (TM - 2) * TN + TN / 2 + 1) a_rd[q] ^= Cfg::A_IMG;    // (behind the last A read of this stage)
            if constexpr (t == TILES - 3) DGA_BX_READ(Bh[0][q], b_rd[q], 0);
            if constexpr (t == TILES - 2) DGA_BX_READ(Bh[1][q], b_rd[q], 4096);
            if constexpr (t == TILES - 1) DGA_BX_READ(As[0][q], a_rd[q], 0);
            // the next block's scales
            if constexpr (t == TN + 2) s_nxt[q] = load_sfa(q, kb_scale);
            if constexpr (u == 4 * (TN + 3)) sfb_nxt = load_sfb(kb_scale);
            // promotion of tile t - 2, one accumulator element per gap (the first two tiles promote the previous block's last)
            {
                constexpr int j = t >= 2 ? t - 2 : TILES - 2 + t, jm = j / TN, jn = j % TN;
                const float sv = t >= 2 ? s_cur[jm] : s_old3;
#ifdef DGA_BXI_NOFMA    // diagnostic: no promotion
                asm volatile("" ::"v"(part[j & 3][q]), "v"(sv));
#else
                acc[jm][jn][q] = __builtin_fmaf(part[j & 3][q], sv, acc[jm][jn][q]);
#endif
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        DGA_STAMP(4);
        s_old3 = s_cur[3];
#pragma unroll
        for (int i = 0; i < TM; ++i) s_cur[i] = s_nxt[i] * sfb_nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads issued for a block that does not exist
    // drain: the last block's last two tiles
#pragma unroll
    for (int j = TILES - 2; j < TILES; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j / TN][j % TN][q] = __builtin_fmaf(part[j & 3][q], s_old3, acc[j / TN][j % TN][q]);
    DGA_STAMP_CLOCK(6, 7);
    DGA_STAMP_FLUSH();
    loop_clock.tick();
    loop_clock.flush(p.stamps, blockIdx.x * (NT / 64) + wave, lane);
#undef DGA_BX_READ
#undef DGA_BX_READ_ASM

    // ---- epilogue: lane owns row m, columns n_base + 32 j + [0, 8) (two n-tiles = 8 consecutive bf16 = one 16-byte store)
    const int m_row = m0 + wm * WROWS + li;
    const int n_base = n0 + wn * WCOLS + 8 * kg;
    if (p.splitk > 1) {
        float *slab = p.partial + (int64_t)split * p.m * p.n;
        const bool v_ok = (p.n & 3) == 0;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m_row + mt * 16;
            if (m >= M) continue;
            float *prow = slab + (int64_t)m * p.n;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int n = n_base + 32 * (nt >> 1) + 4 * (nt & 1);
                if (v_ok && n + 4 <= p.n) {
                    *(v4f *)(prow + n) = acc[mt][nt];
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < p.n) prow[n + q] = acc[mt][nt][q];
                }
            }
        }
        return;
    }
    const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = m_row + mt * 16;
        if (m >= M) continue;
        uint16_t *crow = C + (int64_t)m * p.ldc;
#pragma unroll
        for (int j = 0; j < TN / 2; ++j) {
            const int n = n_base + 32 * j;
            const v4f lo = acc[mt][2 * j], hi = acc[mt][2 * j + 1];
            const v2bf h0 = __builtin_convertvector(v2f{lo.x, lo.y}, v2bf);
            const v2bf h1 = __builtin_convertvector(v2f{lo.z, lo.w}, v2bf);
            const v2bf h2 = __builtin_convertvector(v2f{hi.x, hi.y}, v2bf);
            const v2bf h3 = __builtin_convertvector(v2f{hi.z, hi.w}, v2bf);
            const v4i pk = v4i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1),
                               __builtin_bit_cast(int, h2), __builtin_bit_cast(int, h3)};
            if (vec_ok && n + 8 <= p.n) {
                *(v4i *)(crow + n) = pk;
            } else {
                const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (n + q < p.n) crow[n + q] = e[q];
            }
        }
    }
}

}  // namespace dga
