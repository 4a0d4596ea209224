// The bf16-exact policy's kernel for the masked grouped layout (dispatchPolicyTag 7, 128 x 256 tile on 2 x 4 waves): a persistent
// kernel like gemm_fp8_bf16x_persistent_kernel.hpp, rebuilt around the two things that bound the expert weight stream there
// (profiles/r06_grouped_masks_base.txt: 5.2-5.4 TB/s with few rows per expert against the fast policy's 6.6-7.0; 879 us on random masks).
//
//  (1) Rows that do not exist are not multiplied, at the granularity of the matrix instruction: a wave whose 64-row share of an expert
//      holds L = 0..4 m-tiles (16 rows each) with rows runs the loop that is unrolled for exactly L -- L x 16 MFMAs, L x 16 A
//      conversions, L x 16 promotions per k block instead of 64 of each.  L is fixed for a whole tile (masked_m[g] is), so the choice
//      is one scalar branch per tile, outside the k loop.  This is the reference's walk over the blocks that exist
//      (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:189-200: the last part multiplies r_m_blocks, not
//      m_sec_o_blocks).
//  (2) TWO k blocks of the ring are in flight, not one.  The persistent kernel converts the next block's fragments in place while it
//      multiplies the current one, which needs the next block LANDED at the top of a block: of three stages one is consumed, one
//      readable and only one in flight, and a stage's round trip (~2.3 us under load) bounds the stream at ~48 KB per 2.3 us per CU.
//      Here a block reads its fragments from its own stage only (first fragments converted in a burst behind the barrier; the burst of
//      one wave overlaps the other wave of its SIMD), so the wait at the top of a block leaves the younger stage's pieces in flight
//      (counted vmcnt, also across the previous tile's output stores).
//
// (Tried on top of (2), not kept: the XCD's L2 as further stages of the ring -- one dword of every 64 bytes of the weight block two or
//  four blocks ahead of the refill, "loaded" by LDS-DMA into a dump slot, so that the refill would hit the L2.  0..16 rows per expert
//  544 -> 692 us two blocks ahead, 897 four: 64 separate lines per wave-instruction cost the memory path more than the refill's round
//  trip did.  profiles/r06_grouped_masks.txt, "L2 prefetch" rows.)
//
// (Tried on the lone tiles, not kept: the idle waves 4..7 converting the tile's A rows once per k block into a bf16 image -- in the upper
//  half of the block's own A stage, which a lone tile never reads, and 8 KB behind the ring -- handed over by a second barrier, so that
//  the multiplying waves convert B only.  Same bits; 64 rows per expert 664 -> 716 us, 0..64 rows 576 -> 612, random masks level: a
//  multiplying wave alone on its SIMD waits out the image's round trip at that barrier in every block, where its own A conversions were
//  hidden in the MFMA gaps.  The decode kernel's image pays because there BOTH waves of a SIMD convert the same rows.)
//
// Same arithmetic in the same k order as every other build of the policy (four chained v_mfma_f32_16x16x32_bf16 per scale block on
// exactly converted operands, one fp32 promotion per block): bit-identical outputs (tests/test_bf16x_grouped_gpu.py).  Masked grouped
// rasters, packed or indexed rows, with K of at least two k blocks; everything else keeps the other builds.
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

template <bool KTAIL, bool BNT, bool STAGGER = true, bool INDEXED = false>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_fp8_bf16x_grouped_kernel(const GemmParams p)
{
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN, TM = Cfg::TM, TN = Cfg::TN, DNT = Cfg::DNT;
    constexpr int NL = Cfg::LOADS_PER_STAGE, LAGT = 2, RING = 4, NT = Cfg::NT;
    static_assert(Cfg::NT == 512 && DNT == 512 && TM == 4 && TN == 4 && Cfg::SC_ITERS == 1 && Cfg::STAGES == 3, "the grouped schedule");
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kg = lane >> 4;
    const int KB = p.kb_n;   // >= 2 (host)

    // ---- this workgroup's tile list (gemm_fp8_persistent_kernel.hpp): one contiguous chunk of the raster per XCD
    const int tiles_per_group = p.tiles_m * p.tiles_n;
    const int total = p.groups * tiles_per_group;
    int first = 0, count = total, step = gridDim.x, slot = blockIdx.x;
    if (p.xcd_remap) {
        const int xcd = blockIdx.x & 7, q = total >> 3, r = total & 7;
        first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        count = q + (xcd < r ? 1 : 0);
        step = ((int)gridDim.x - xcd + 7) >> 3;
        slot = blockIdx.x >> 3;
    }
    struct Tile { int g, M, m0, n0; };
    typedef const __attribute__((address_space(4))) int32_t *const_i32_ptr;   // scalar loads (see gemm_fp8_persistent_kernel.hpp)
    const const_i32_ptr masked_m_c = (const_i32_ptr)p.masked_m;
    auto seek = [&](int &local, Tile &t) -> bool {
        for (; local < count; local += step) {
            const int tile = first + local;
            const int g = tile / tiles_per_group;
            const int t_in = tile - g * tiles_per_group;
            const int gm = p.raster_group;
            const int per = gm * p.tiles_n;
            const int band = t_in / per;
            const int row0 = band * gm;
            const int rows = min(p.tiles_m - row0, gm);
            const int loc = t_in - band * per;
            const int tm = row0 + loc % rows, tn = loc / rows;
            const int M = p.masked_m ? min(masked_m_c[g], p.m) : p.m;
            const int m0 = tm * BM;
            if (m0 >= M) continue;              // empty expert / fully masked tile
            t.g = g; t.M = M; t.m0 = m0; t.n0 = tn * BN;
            return true;
        }
        return false;
    };

    // ---- the FILL tile: the tile whose k blocks the refill fetches (two blocks ahead of the block being multiplied, so during a
    //      tile's last two blocks it is the next tile).  Its descriptors and per-lane offsets are the only DMA state there is.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    constexpr uint32_t kOutOfRange = 0x80000000u;
    const int a_col = ((tid & 7) ^ swz_a(tid >> 3)) * 16;
    const int b_col = ((tid & 7) ^ swz_b(tid >> 3)) * 16;
    // A wave issues its own pieces (thread slot tid: offsets kept in registers) and -- a wave of the second row that has no rows to
    // multiply, see `lone` below -- IN ITS PARTNER'S PLACE the pieces of the wave four below it (slot tid - 256: offsets computed
    // where they are used, by a wave that has nothing else to do; kept, they cost the registers the L = 4 loop needs).
    v4i a_rsrc, b_rsrc;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
    const float *sc_src;
    int kbf = 0;            // the fill tile's next k block
    bool fill_valid = true;
    // Indexed form (GemmParams::row_index; dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed): row r of expert g is row
    // row_index[g * m + r] of ONE flat source (group strides 0); its scales and its result row are found the same way.  The index
    // entries are plain loads where a tile becomes the fill tile / is stored -- the compiler drains the wave's refills in front of
    // their first use, once per tile; a lone tile's idle waves therefore KEEP their partners' offsets in this form instead of computing
    // them per piece (that would drain the refill in every block): 16 bytes per thread of waves 4..7 behind the ring in LDS, written
    // with the fill tile and read back where a piece is issued (in registers they spill the L = 4 loop: 256 + 9).
    constexpr bool indexed = INDEXED;     // (a build of its own: as a runtime flag it takes the packed build from 245 to 256 registers)
    static_assert(Cfg::A_ITERS == 2, "partner slot: two A offsets and the scale pointer");
    uint32_t *const pslot = reinterpret_cast<uint32_t *>(smem + Cfg::LDS_BYTES) + (tid & (DNT / 2 - 1)) * 4;
    auto a_off_of = [&](const Tile &t, int it, int td) -> uint32_t {
        const int row = (it * DNT + td) >> 3;
        if (row >= t.M - t.m0) return kOutOfRange;                                          // rows at or beyond M: zero-filled, not fetched
        const uint32_t r = indexed ? (uint32_t)p.row_index[(int64_t)t.g * p.m + t.m0 + row] : (uint32_t)row;
        return r * (uint32_t)p.lda + a_col;
    };
    auto b_off_of = [&](const Tile &t, int it, int td) -> uint32_t {
        const int row = (it * DNT + td) >> 3;
        return (uint32_t)min(row, p.n - 1 - t.n0) * (uint32_t)p.ldb + b_col;
    };
    auto sc_of = [&](const Tile &t, int td) -> const float * {   // slot td: [0, BM) sfa rows of the tile, then its sfb blocks
        const float *SFA = p.sfa + (int64_t)t.g * p.sfa_gs, *SFB = p.sfb + (int64_t)t.g * p.sfb_gs;
        if (td >= BM) return SFB + (int64_t)min(t.n0 / 128 + min(td - BM, 7), p.nb_n - 1) * p.kb_n;
        const int mr = min(t.m0 + td, t.M - 1);
        return SFA + (indexed ? p.row_index[(int64_t)t.g * p.m + mr] : (int64_t)mr) * p.sfa_ld;
    };
    Tile T{}, F{};
    auto set_fill = [&](const Tile &t) {
        a_rsrc = indexed ? make_rsrc(p.a, p.a_bytes)
                         : make_rsrc(p.a + (int64_t)t.g * p.a_gs + (int64_t)t.m0 * p.lda, (int64_t)(t.M - t.m0) * p.lda);
        b_rsrc = make_rsrc(p.b + (int64_t)t.g * p.b_gs + (int64_t)t.n0 * p.ldb, (int64_t)(p.n - t.n0) * p.ldb);
#pragma unroll
        for (int it = 0; it < Cfg::A_ITERS; ++it) a_voff[it] = a_off_of(t, it, tid);
#pragma unroll
        for (int it = 0; it < Cfg::B_ITERS; ++it) b_voff[it] = b_off_of(t, it, tid);
        sc_src = sc_of(t, tid);
        if constexpr (INDEXED) {
            if (wm == 1) {     // (needed while the tile being multiplied is lone, whatever the fill tile is)
                const uint64_t ps = (uint64_t)(uintptr_t)sc_of(t, tid - DNT / 2);
                *reinterpret_cast<v4i *>(pslot) = v4i{(int)a_off_of(t, 0, tid - DNT / 2), (int)a_off_of(t, 1, tid - DNT / 2), (int)(uint32_t)ps, (int)(uint32_t)(ps >> 32)};
            }
        }
    };
    // piece idx of stage `stage` from the fill tile's k block kbf, for this wave (S = 0) or in its partner's place (S = 1).  No fill
    // tile: every lane out of range -- zeros land, nothing is fetched; the scale piece re-reads a block of the last tile
    auto refill = [&](int idx, int stage, auto sc) {
        constexpr int S = decltype(sc)::value;
        const int w = wave - 4 * S, td = tid - S * (DNT / 2);
        const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES + w * 1024;
        const int k0 = kbf * 128;
        if (idx < Cfg::A_ITERS) {
            uint32_t voff = fill_valid ? (S ? (INDEXED ? pslot[idx] : a_off_of(F, idx, td)) : a_voff[idx]) : kOutOfRange;
            if constexpr (KTAIL) voff = (k0 + a_col < p.k) ? voff : kOutOfRange;
            dma16(voff, a_rsrc, (uint32_t)k0, sa + idx * DNT * 16);
        } else if (idx < Cfg::A_ITERS + Cfg::B_ITERS) {
            const int it = idx - Cfg::A_ITERS;
            uint32_t voff = fill_valid ? (S ? b_off_of(F, it, td) : b_voff[it]) : kOutOfRange;
            if constexpr (KTAIL) voff = (k0 + b_col < p.k) ? voff : kOutOfRange;
            if constexpr (BNT) dma16_nt(voff, b_rsrc, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
            else dma16(voff, b_rsrc, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
        } else {
            const float *src = S ? (INDEXED ? reinterpret_cast<const float *>((uintptr_t)*reinterpret_cast<const uint64_t *>(pslot + 2)) : sc_of(F, td)) : sc_src;
            dma4(src + (fill_valid ? kbf : KB - 1), lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES + w * 256);
        }
    };
    constexpr std::integral_constant<int, 0> kOwn{};
    constexpr std::integral_constant<int, 1> kPartner{};

    // ---- per-lane fragment read offsets (bytes inside a stage): gemm_fp8_kernel.hpp
    const int a_row = wm * (BM / Cfg::kWM) + li;
    const int a_off0 = a_row * 128 + ((kg ^ swz_a(a_row)) * 16);
    const int a_off1 = a_row * 128 + (((kg + 4) ^ swz_a(a_row)) * 16);
    const int b_row = wn * (BN / WN) + 8 * (li >> 2) + (li & 3);
    const int b_off0 = Cfg::A_BYTES + b_row * 128 + ((kg ^ swz_b(b_row)) * 16);
    const int b_off1 = Cfg::A_BYTES + b_row * 128 + (((kg + 4) ^ swz_b(b_row)) * 16);
    const int sa_off = Cfg::A_BYTES + Cfg::B_BYTES + (wm * (BM / Cfg::kWM) + li) * 4;
    const int sb_off = Cfg::A_BYTES + Cfg::B_BYTES + (BM + (wn * (BN / WN)) / 128) * 4;

    v4f acc[TM][TN];
    v4f part[RING];
    v4i afx[TM][4], bfx[2][4];      // bf16 fragments: [q] = the 8 bf16 of MFMA q of the chain
    v4i braw[2], araw[2][2];         // raw e4m3 bytes: [0] = bytes [16 kg, +16), [1] = bytes [64 + 16 kg, +16)
    float s_cur[TM], s_old[TM];
    auto convert = [](const v4i (&raw)[2], v4i (&dst)[4], int c) {
        const int w = raw[(c >> 1) >> 2][(c >> 1) & 3];
        dst[c >> 2][c & 3] = (c & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                                     : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
    };
    auto b_frag_off = [](int nt) { return (nt >> 1) * 4096 + (nt & 1) * 512; };

    // counted wait at the top of a block: everything but the NL pieces of the younger stage in flight -- and, in a tile's first two
    // blocks, but the previous tile's output stores, which were issued behind those pieces (never fewer stores than `extra`)
    auto wait_all_but = [&](auto base, int extra) {
        constexpr int B = decltype(base)::value;
        if (extra >= 8) wait_vmcnt<B + 8>();
        else if (extra >= 6) wait_vmcnt<B + 6>();
        else if (extra >= 4) wait_vmcnt<B + 4>();
        else if (extra >= 2) wait_vmcnt<B + 2>();
        else wait_vmcnt<B>();
    };
    // `younger` = the pieces this wave issued during the previous block (0, NL or 2 NL: see `lone`)
    auto wait_landed = [&](int younger, int extra) {
        if (younger >= 2 * NL) wait_all_but(std::integral_constant<int, 2 * NL>{}, extra);
        else if (younger >= NL) wait_all_but(std::integral_constant<int, NL>{}, extra);
        else wait_all_but(std::integral_constant<int, 0>{}, extra);
    };

    int local = slot;
    if (!seek(local, T)) return;
    F = T;
    set_fill(F);
    // ---- prologue: blocks 0 and 1 of the first tile on their way
#pragma unroll
    for (int d = 0; d < 2; ++d) {
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) refill(idx, d, kOwn);
        ++kbf;
    }
    int cur = 0;             // the stage of the block being multiplied; its predecessor's stage, (cur + 2) % 3, is refilled
    DGA_STAMP_DECL           // (diagnostic builds only: scripts/ubench/stamp_grouped_bx.hip)
    DGA_STAMP_CLOCK(6, 7);
    DGA_STAMP_START();
    int stores_pending = 0;  // output stores of the previous tile issued by this wave (a lower bound)
    int younger = NL;        // pieces this wave issued during the previous block

    // the fill tile moves on when its last block has been issued
    auto advance_fill = [&]() {
        if (kbf == KB) {
            local += step;
            fill_valid = seek(local, F);
            if (fill_valid) set_fill(F);
            kbf = 0;
        }
    };

    // one tile with L m-tiles of this wave's 64 rows present (L = 0: the wave only takes part in the refill and the barriers).
    // STAG: the wave runs HALF A BLOCK behind the barrier -- between two barriers it finishes the previous block (n-tiles 2 and 3:
    // registers only, no LDS read) and then reads and starts the block the barrier announced.  The two waves of a SIMD then do not
    // stand in their first-fragment bursts at the same time (the port idles there: profiles/r06_grouped_stamps.txt, head 880 / 2400
    // ticks), one converts while the other multiplies (MI355X_MICROARCH.md "Two waves per SIMD" item 9: stagger waves 4..7).
    auto run_tile = [&](auto Lc, auto lonec, auto stagc) __attribute__((always_inline)) {
        constexpr int L = decltype(Lc)::value;
        constexpr bool lone = decltype(lonec)::value;
        constexpr bool STAG = decltype(stagc)::value;
        constexpr int G = 4 * (L > 0 ? L : 1);       // MFMA gaps per n-tile
        constexpr int TILES = (L > 0 ? L : 1) * TN;
        constexpr int SP = L > 0 ? L : 1;            // gaps between two refill pieces
        constexpr int HALF = 2 * TILES;              // gaps of n-tiles 0 and 1
        static_assert(1 + (NL - 1) * SP < HALF, "the refill fits half a block");
        static_assert(!(STAG && (lone || L == 0)), "a staggered wave multiplies and issues its own pieces");
        if constexpr (L > 0) {
#pragma unroll
            for (int i = 0; i < L; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < L; ++i) s_old[i] = 0.f;    // the first LAGT tiles "promote the previous block": part (= 0) * 0
        }
        // the block's first fragments out of its stage: A(0), B(0) converted in a burst
        auto head = [&](const uint8_t *sc) __attribute__((always_inline)) {
            araw[0][0] = *(const v4i *)(sc + a_off0);
            araw[0][1] = *(const v4i *)(sc + a_off1);
            braw[0] = *(const v4i *)(sc + b_off0);
            braw[1] = *(const v4i *)(sc + b_off1);
            if constexpr (L > 1) {
                araw[1][0] = *(const v4i *)(sc + a_off0 + 2048);
                araw[1][1] = *(const v4i *)(sc + a_off1 + 2048);
            }
            const float sfb0 = *(const float *)(sc + sb_off);
            float sa[L > 0 ? L : 1];
#pragma unroll
            for (int i = 0; i < L; ++i) sa[i] = *(const float *)(sc + sa_off + i * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 8; ++c) convert(braw, bfx[0], c);
            braw[0] = *(const v4i *)(sc + b_off0 + b_frag_off(1));    // B(1), raw
#pragma unroll
            for (int c = 8; c < 16; ++c) convert(braw, bfx[0], c);
            braw[1] = *(const v4i *)(sc + b_off1 + b_frag_off(1));
#pragma unroll
            for (int c = 0; c < 16; ++c) convert(araw[0], afx[0], c);
            if constexpr (L > 2) {
                araw[0][0] = *(const v4i *)(sc + a_off0 + 2 * 2048);
                araw[0][1] = *(const v4i *)(sc + a_off1 + 2 * 2048);
            }
#pragma unroll
            for (int i = 0; i < L; ++i) s_cur[i] = sa[i] * sfb0;
            __builtin_amdgcn_sched_barrier(0);
        };
        // MFMA gaps [U0, U1) of a block; REFILL: the refill rides on them, one piece per SP gaps from the second gap on.  (Issued in a
        // burst behind the barrier the 56 pieces of the workgroup queue on the CU's one vector-memory path and every wave's first MFMA
        // waits for the last of them.)  Gaps of n-tiles 2 and 3 read no LDS.
        auto gaps = [&](auto u0c, auto u1c, auto refillc, const uint8_t *sc, int fill) __attribute__((always_inline)) {
            constexpr int U0 = decltype(u0c)::value, U1 = decltype(u1c)::value;
            constexpr bool REFILL = decltype(refillc)::value;
#pragma unroll
            for (int u = U0; u < U1; ++u) {
                const int t = u >> 2, q = u & 3, nt = t / L, mt = t % L, g = u % G;
                part[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(v8bf, bfx[nt & 1][q]), __builtin_bit_cast(v8bf, afx[mt][q]),
                    q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t % RING], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (REFILL && u - U0 >= 1 && (u - U0 - 1) % SP == 0 && (u - U0 - 1) / SP < NL) refill((u - U0 - 1) / SP, fill, kOwn);
                // A(mt + 1) is converted behind the MFMAs of the first n-tile's m-tile mt; its raw bytes were read a tile earlier
                if (nt == 0 && mt + 1 < L) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(mt + 1) & 1], afx[mt + 1], 4 * q + c);
                    if (q == 3 && mt + 3 < L) {
                        araw[(mt + 1) & 1][0] = *(const v4i *)(sc + a_off0 + (mt + 3) * 2048);
                        araw[(mt + 1) & 1][1] = *(const v4i *)(sc + a_off1 + (mt + 3) * 2048);
                    }
                }
                // B(nt + 1) is converted behind the MFMAs of n-tile nt into the other bf16 set; the raw halves are reloaded for
                // B(nt + 2) as the conversions release them
                if (nt + 1 < TN) {
#pragma unroll
                    for (int c = 16 * g / G; c < 16 * (g + 1) / G; ++c) convert(braw, bfx[(nt + 1) & 1], c);
                    if (nt + 2 < TN) {
                        if (g == G / 2 - 1) braw[0] = *(const v4i *)(sc + b_off0 + b_frag_off(nt + 2));
                        if (g == G - 1) braw[1] = *(const v4i *)(sc + b_off1 + b_frag_off(nt + 2));
                    }
                }
                {   // the promotion of tile t - LAGT (the previous block's last tiles during this block's first ones)
                    const int j = t >= LAGT ? t - LAGT : TILES + t - LAGT, jn = j / L, jm = j % L;
                    const float sv = t >= LAGT ? s_cur[jm] : s_old[jm];
                    acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], sv, acc[jm][jn][q]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        constexpr std::integral_constant<int, 0> kU0{};
        constexpr std::integral_constant<int, HALF> kUh{};
        constexpr std::integral_constant<int, 4 * TILES> kU1{};
        auto end_of_block = [&]() {
#pragma unroll
            for (int i = 0; i < L; ++i) s_old[i] = s_cur[i];
        };
        // the top of a block: this wave's pieces of it have landed; everyone's have; and everyone has left the block whose stage is
        // refilled now.  Returns that stage.
        auto top = [&](int kb) __attribute__((always_inline)) -> int {
            DGA_STAMP(0);
            wait_landed(younger, kb < 2 ? stores_pending : 0);
            DGA_STAMP(1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            DGA_STAMP(2);
            advance_fill();
            return cur >= 1 ? cur - 1 : 2;
        };
        auto next_block = [&]() {
            ++kbf;
            cur = cur == 2 ? 0 : cur + 1;
        };
        if constexpr (STAG) {
            // (a static priority for this half -- s_setprio 1 around its tile -- measured 2-3 % SLOWER at 80 / 96 rows: not done)
            // block 0: nothing to finish -- the refill in a burst, the head, n-tiles 0 and 1
            {
                const int fill = top(0);
                younger = NL;
#pragma unroll
                for (int idx = 0; idx < NL; ++idx) refill(idx, fill, kOwn);
                const uint8_t *sc = smem + cur * Cfg::STAGE_BYTES;
                head(sc);
                gaps(kU0, kUh, std::false_type{}, sc, fill);
                next_block();
            }
            for (int kb = 1; kb < KB; ++kb) {
                const int fill = top(kb);
                const uint8_t *sc = smem + cur * Cfg::STAGE_BYTES;
                gaps(kUh, kU1, std::true_type{}, sc, fill);     // the previous block's n-tiles 2 and 3 (no LDS read), the refill on them
                end_of_block();
                head(sc);
                DGA_STAMP(3);
                gaps(kU0, kUh, std::false_type{}, sc, fill);
                DGA_STAMP(4);
                next_block();
            }
            gaps(kUh, kU1, std::false_type{}, smem, 0);         // the last block's second half
            end_of_block();
        } else {
            for (int kb = 0; kb < KB; ++kb) {
                const int fill = top(kb);
                if constexpr (L == 0) {     // (a wave of the second row, in a lone tile)
#pragma unroll
                    for (int idx = 0; idx < NL; ++idx) {
                        if (lone) refill(idx, fill, kPartner);
                        refill(idx, fill, kOwn);
                    }
                    younger = lone ? 2 * NL : NL;
                } else {
                    younger = lone ? 0 : NL;
                    const uint8_t *sc = smem + cur * Cfg::STAGE_BYTES;
                    head(sc);
                    DGA_STAMP(3);
                    if constexpr (lone) gaps(kU0, kU1, std::false_type{}, sc, fill);
                    else gaps(kU0, kU1, std::true_type{}, sc, fill);
                    end_of_block();
                    DGA_STAMP(4);
                }
                next_block();
            }
        }
        stores_pending = 0;
        if constexpr (L > 0) {
            // ---- boundary: the last LAGT tiles of the last block, the stores
#pragma unroll
            for (int t = 0; t < LAGT; ++t) {
                const int j = TILES + t - LAGT, jn = j / L, jm = j % L;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], s_old[jm], acc[jm][jn][q]);
            }
            uint16_t *C = p.out + (int64_t)T.g * p.c_gs;
            const int m_row = T.m0 + wm * (BM / Cfg::kWM) + li;
            const int n_base = T.n0 + wn * (BN / WN) + 8 * kg;
            const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
            // every m-tile below L has a row, so each of the 2 L vector stores below is issued (full-width tiles of aligned rows)
            if (vec_ok && T.n0 + BN <= p.n) stores_pending = 2 * L;
#pragma unroll
            for (int mt = 0; mt < L; ++mt) {
                const int m = m_row + mt * 16;
                if (m >= T.M) continue;
                uint16_t *crow = C + (indexed ? p.row_index[(int64_t)T.g * p.m + m] : (int64_t)m) * p.ldc;
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
                        if (p.out_nt == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(crow + n), "v"(pk) : "memory");
                        else if (p.out_nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(crow + n), "v"(pk) : "memory");
                        else if (p.out_nt == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(crow + n), "v"(pk) : "memory");
                        else asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(crow + n), "v"(pk) : "memory");
                    } else {
                        const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (n + q < p.n) crow[n + q] = e[q];
                    }
                }
            }
        }
    };

    for (;;) {
        // m-tiles of this wave's rows that exist in this tile (wave-uniform, fixed for the tile)
        const int rows = T.M - T.m0 - wm * (BM / Cfg::kWM);
        const int L = rows <= 0 ? 0 : (rows >= 64 ? 4 : (rows + 15) >> 4);
        // A tile whose rows all lie in the first wave row's half (at most 64 rows: waves 4..7 multiply nothing) has the idle waves
        // issue the whole refill, their partners' pieces too: a wave alone on its SIMD pays every piece it issues with ~50 cycles of
        // its own MFMA stream (profiles/r06_grouped_stamps.txt), an idle wave pays nothing.  The role holds for a tile (it is a
        // compile-time property of the tile's loop: a branch per MFMA gap costs the loop its registers); the counted wait at the top
        // of a block goes by what THIS wave issued a block ago.
        const bool lone = T.M - T.m0 <= BM / Cfg::kWM;
        constexpr std::true_type yes{};
        constexpr std::false_type no{};
        if (lone) {
            if (L == 0) run_tile(std::integral_constant<int, 0>{}, yes, no);
            else if (L == 1) run_tile(std::integral_constant<int, 1>{}, yes, no);
            else if (L == 2) run_tile(std::integral_constant<int, 2>{}, yes, no);
            else if (L == 3) run_tile(std::integral_constant<int, 3>{}, yes, no);
            else run_tile(std::integral_constant<int, 4>{}, yes, no);
        } else if (wm == 0) {                       // both wave rows multiply: the first one has all four m-tiles ...
            run_tile(std::integral_constant<int, 4>{}, no, no);
        } else {                                    // ... the second one runs half a block behind it
            constexpr std::integral_constant<bool, STAGGER> st{};
            if (L == 1) run_tile(std::integral_constant<int, 1>{}, no, st);
            else if (L == 2) run_tile(std::integral_constant<int, 2>{}, no, st);
            else if (L == 3) run_tile(std::integral_constant<int, 3>{}, no, st);
            else run_tile(std::integral_constant<int, 4>{}, no, st);
        }
        // the tile after this one is the fill tile (the fill moved on during this tile's last two blocks)
        if (!fill_valid) break;
        T = F;
    }
    DGA_STAMP(5);
    DGA_STAMP_CLOCK(6, 7);
    DGA_STAMP_FLUSH();
    wait_vmcnt<0>();   // the refills past the last tile (zeros) land in LDS nobody reads: drain them before exit
}

}  // namespace dga
