// Persistent form of the bf16-exact policy's 128 x 256 in-register build (gemm_fp8_kernel.hpp, MATH = 1; dispatchPolicyTag 7): one
// workgroup per CU walks a strided list of tiles and the LDS ring does not stop at a tile boundary.
//
// Why.  With one workgroup per tile every tile pays an un-overlapped start (descriptor set-up, the first two stages' round trip with
// nothing in flight, the first fragments converted in one burst) and an un-overlapped end (the bf16 store burst with an empty DMA
// queue): ~10 % of the launch on the masked grouped stream (8 tiles per CU), ~8 of 112 us at 4096^3 (2 tiles per CU).  Here the
// refill slots of a tile's last two k blocks fetch the NEXT tile's first two blocks, the in-place fragment conversions at the end
// of the last k block pick up the next tile's block 0 like any other block, and the boundary is the drain of two promotions, the
// stores, and the clearing of the accumulators (the promotion ring too: a NaN left there would meet the zero scale of the next
// tile's first steps).  The same construction as gemm_fp8_cont_persistent_kernel.hpp has for the fast policy's 256 x 256 tile.
//
// Same arithmetic in the same order as the one-tile build: bit-identical (tests/test_bf16_exact_gpu.py, test_bf16x_persistent_gpu.py).
// Dense and masked grouped rasters; K of at least two k blocks; no split-K, no indexed rows, no contiguous layout (those keep the
// one-tile build).  A wave whose rows all lie at or beyond M multiplies nothing in that tile (it keeps its share of the refill DMA
// and the barriers); when its next tile has rows again it sets its fragments up from the stage that tile's block 0 has landed in.
// (Tried and dropped: per-wave loop variants that multiply only the 1, 2 or 4 m-tiles holding rows.  Same bits, but with four variants
//  of the 64-gap body in one function the full-schedule loop -- 239 of 256 VGPRs on its own -- lost 13 % (4096^3 120 -> 136 us) and the
//  short ones did not reach the tile-height hint of dga_tiling_bf16_exact (0..16 rows per expert: 773 us against 617).)
// Counterpart in the reference: its kernel is persistent by construction -- one block per AI core walks the tiles of its section
// with double-buffered L1 across them (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:160-198).
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

template <bool KTAIL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_fp8_bf16x_persistent_kernel(const GemmParams p)
{
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN, TM = Cfg::TM, TN = Cfg::TN, DNT = Cfg::DNT;
    constexpr int NL = Cfg::LOADS_PER_STAGE, TILES = TM * TN, G = 4 * TM, LAGT = 2, RING = 4;
    static_assert(Cfg::NT == 512 && DNT == 512 && TILES % RING == 0 && 4 * TILES >= 4 + NL && 16 % G == 0, "the MATH = 1 schedule");
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kg = lane >> 4;
    const int KB = p.kb_n;   // >= 2 (host)

    // ---- this workgroup's tile list (as gemm_fp8_persistent_kernel.hpp): one contiguous chunk of the raster per XCD, walked
    //      together by the workgroups of that XCD, `step` tiles per round
    const int tiles_per_group = p.tiles_m * p.tiles_n;
    const int total = p.launch_tiles > 0 ? p.launch_tiles : p.groups * tiles_per_group;   // (launch_tiles: the first tiles of a dense raster, the rest follows in another launch)
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

    // ---- LDS-DMA sources.  The tile being multiplied keeps its per-lane offsets in registers (every k block uses them); the
    //      next tile's are computed where they are used -- fourteen instructions per tile, in its predecessor's last two k blocks.
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    constexpr uint32_t kOutOfRange = 0x80000000u;
    const int a_col = ((tid & 7) ^ swz_a(tid >> 3)) * 16;
    const int b_col = ((tid & 7) ^ swz_b(tid >> 3)) * 16;
    auto a_off = [&](const Tile &t, int it) -> uint32_t {
        const int row = (it * DNT + tid) >> 3;
        return row < t.M - t.m0 ? (uint32_t)row * (uint32_t)p.lda + a_col : kOutOfRange;   // rows at or beyond M: zero-filled, not fetched
    };
    auto b_off = [&](const Tile &t, int it) -> uint32_t {
        const int row = (it * DNT + tid) >> 3;
        return (uint32_t)min(row, p.n - 1 - t.n0) * (uint32_t)p.ldb + b_col;
    };
    auto sc_ptr = [&](const Tile &t) -> const float * {   // slot tid: [0, BM) sfa rows of the tile, then its sfb blocks
        const float *SFA = p.sfa + (int64_t)t.g * p.sfa_gs, *SFB = p.sfb + (int64_t)t.g * p.sfb_gs;
        return tid < BM ? SFA + (int64_t)min(t.m0 + tid, t.M - 1) * p.sfa_ld
                        : SFB + (int64_t)min(t.n0 / 128 + min(tid - BM, 7), p.nb_n - 1) * p.kb_n;
    };
    auto a_desc = [&](const Tile &t) { return make_rsrc(p.a + (int64_t)t.g * p.a_gs + (int64_t)t.m0 * p.lda, (int64_t)(t.M - t.m0) * p.lda); };
    auto b_desc = [&](const Tile &t) { return make_rsrc(p.b + (int64_t)t.g * p.b_gs + (int64_t)t.n0 * p.ldb, (int64_t)(p.n - t.n0) * p.ldb); };
    static_assert(Cfg::SC_ITERS == 1, "one scale piece per stage");

    Tile T{}, Tn{};
    int local = slot;
    if (!seek(local, T)) return;
    v4i a_rsrc = a_desc(T), b_rsrc = b_desc(T), a_rsrc_n = a_rsrc, b_rsrc_n = b_rsrc;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
    const float *sc_src;
    auto load_offsets = [&](const Tile &t) {
#pragma unroll
        for (int it = 0; it < Cfg::A_ITERS; ++it) a_voff[it] = a_off(t, it);
#pragma unroll
        for (int it = 0; it < Cfg::B_ITERS; ++it) b_voff[it] = b_off(t, it);
        sc_src = sc_ptr(t);
    };
    load_offsets(T);
    // piece idx of a stage, from the current tile's k block kb
    auto issue_cur = [&](int idx, int stage, int kb) {
        const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES + wave * 1024;
        const int k0 = kb * 128;
        if (idx < Cfg::A_ITERS) {
            uint32_t voff = a_voff[idx];
            if constexpr (KTAIL) voff = (k0 + a_col < p.k) ? voff : kOutOfRange;
            dma16(voff, a_rsrc, (uint32_t)k0, sa + idx * DNT * 16);
        } else if (idx < Cfg::A_ITERS + Cfg::B_ITERS) {
            const int it = idx - Cfg::A_ITERS;
            uint32_t voff = b_voff[it];
            if constexpr (KTAIL) voff = (k0 + b_col < p.k) ? voff : kOutOfRange;
            dma16(voff, b_rsrc, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
        } else {
            dma4(sc_src + kb, lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES + wave * 256);
        }
    };
    // the same from the NEXT tile's k block kb (has_next == false: every lane out of range -- zeros land, nothing is fetched; the
    // scale piece re-reads the current tile's last block)
    bool has_next = false;
    auto issue_next = [&](int idx, int stage, int kb) {
        const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES + wave * 1024;
        const int k0 = kb * 128;
        if (idx < Cfg::A_ITERS) {
            uint32_t voff = has_next ? a_off(Tn, idx) : kOutOfRange;
            if constexpr (KTAIL) voff = (k0 + a_col < p.k) ? voff : kOutOfRange;
            dma16(voff, a_rsrc_n, (uint32_t)k0, sa + idx * DNT * 16);
        } else if (idx < Cfg::A_ITERS + Cfg::B_ITERS) {
            const int it = idx - Cfg::A_ITERS;
            uint32_t voff = has_next ? b_off(Tn, it) : kOutOfRange;
            if constexpr (KTAIL) voff = (k0 + b_col < p.k) ? voff : kOutOfRange;
            dma16(voff, b_rsrc_n, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
        } else {
            dma4(has_next ? sc_ptr(Tn) + kb : sc_src + (KB - 1), lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES + wave * 256);
        }
    };

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
    float s_cur[TM], s_old[TM], s_nxt[TM];
    auto clear_tile = [&]() {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < TM; ++i) s_old[i] = 0.f;    // the first LAGT tiles "promote the previous block": part (= 0) * 0
    };
    auto convert = [](const v4i (&raw)[2], v4i (&dst)[4], int c) {
        const int w = raw[(c >> 1) >> 2][(c >> 1) & 3];
        dst[c >> 2][c & 3] = (c & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                                     : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
    };
    auto b_frag_off = [](int nt) { return (nt >> 1) * 4096 + (nt & 1) * 512; };
    // the fragments of a tile's block 0 out of the stage it has landed in, converted in one burst (a tile's first block in this
    // wave: the kernel's first tile, and a tile that follows one in which the wave had no rows)
    auto first_fragments = [&](const uint8_t *st) {
        const float sfb0 = *(const float *)(st + sb_off);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            araw[mt & 1][0] = *(const v4i *)(st + a_off0 + mt * 2048);
            araw[mt & 1][1] = *(const v4i *)(st + a_off1 + mt * 2048);
#pragma unroll
            for (int c = 0; c < 16; ++c) convert(araw[mt & 1], afx[mt], c);
            s_cur[mt] = *(const float *)(st + sa_off + mt * 64) * sfb0;
            s_nxt[mt] = 0.f;
        }
        braw[0] = *(const v4i *)(st + b_off0);
        braw[1] = *(const v4i *)(st + b_off1);
#pragma unroll
        for (int c = 0; c < 16; ++c) convert(braw, bfx[0], c);
        braw[0] = *(const v4i *)(st + b_off0 + b_frag_off(1));   // B(1) of block 0, raw
        braw[1] = *(const v4i *)(st + b_off1 + b_frag_off(1));
    };
    auto rows_present = [&](const Tile &t) { return t.m0 + wm * (BM / Cfg::kWM) < t.M; };   // (wave-uniform)

    // ---- prologue: blocks 0 and 1 of the first tile on their way, block 0 landed
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_cur(idx, d, d);
    wait_vmcnt<NL>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    clear_tile();
    bool active = rows_present(T);
    if (active) first_fragments(smem);
    int cur = 0, nxt = 1, fill = 2;

    // one k block; LAST2: the refill is the next tile's block kb + 2 - KB
    auto k_block = [&](int kb, auto last2c) __attribute__((always_inline)) {
        constexpr bool LAST2 = decltype(last2c)::value;
        auto refill = [&](int idx) {
            if constexpr (LAST2) issue_next(idx, fill, kb + 2 - KB);
            else issue_cur(idx, fill, kb + 2);
        };
        wait_vmcnt<0>();                         // this wave's pieces of the next block (issued a block ago) have landed
        __builtin_amdgcn_s_barrier();            // ... everyone's have; and everyone has left the block whose stage is refilled now
        asm volatile("" ::: "memory");
        if (!active) {
#pragma unroll
            for (int idx = 0; idx < NL; ++idx) refill(idx);
        } else {
            const uint8_t *sc = smem + cur * Cfg::STAGE_BYTES;   // being consumed (B raw reloads of this block)
            const uint8_t *sn = smem + nxt * Cfg::STAGE_BYTES;   // landed: the next block's fragments are read ahead from it
#pragma unroll
            for (int u = 0; u < 4 * TILES; ++u) {
                const int t = u >> 2, q = u & 3, nt = t / TM, mt = t % TM, g = u % G;
                part[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(v8bf, bfx[nt & 1][q]), __builtin_bit_cast(v8bf, afx[mt][q]),
                    q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t % RING], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (u >= 4 && u < 4 + NL) refill(u - 4);
#pragma unroll
                for (int c = 0; c < 16 / G; ++c) convert(braw, bfx[(nt + 1) & 1], (16 / G) * g + c);
                {
                    const int nn = nt + 2;
                    const uint8_t *src = nn < TN ? sc : sn;
                    const int off = b_frag_off(nn < TN ? nn : nn - TN);
                    if (g == G / 2 - 1) braw[0] = *(const v4i *)(src + b_off0 + off);
                    if (g == G - 1) braw[1] = *(const v4i *)(src + b_off1 + off);
                }
                if (nt == TN - 1 && q == 0) {
                    araw[mt & 1][0] = *(const v4i *)(sn + a_off0 + mt * 2048);
                    araw[mt & 1][1] = *(const v4i *)(sn + a_off1 + mt * 2048);
                }
                if (nt == TN - 1 && mt >= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(mt - 1) & 1], afx[mt - 1], 4 * q + c);
                }
                if (t == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(TM - 1) & 1], afx[TM - 1], 4 * q + c);
                }
                if (u == 4 * TILES - 8) {
                    const float sfbn = *(const float *)(sn + sb_off);
#pragma unroll
                    for (int i = 0; i < TM; ++i) s_nxt[i] = *(const float *)(sn + sa_off + i * 64) * sfbn;
                }
                {
                    const int j = t >= LAGT ? t - LAGT : TILES + t - LAGT, jn = j / TM, jm = j % TM;
                    const float sv = t >= LAGT ? s_cur[jm] : s_old[jm];
                    acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], sv, acc[jm][jn][q]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                s_old[i] = s_cur[i];
                s_cur[i] = s_nxt[i];
            }
        }
        const int f = cur;
        cur = nxt; nxt = fill; fill = f;
    };

    for (;;) {
        // the tile after this one (its descriptors are needed from this tile's second-to-last k block on)
        int local_n = local + step;
        has_next = seek(local_n, Tn);
        if (has_next) { a_rsrc_n = a_desc(Tn); b_rsrc_n = b_desc(Tn); }
        for (int kb = 0; kb < KB - 2; ++kb) k_block(kb, std::false_type{});
        k_block(KB - 2, std::true_type{});
        k_block(KB - 1, std::true_type{});

        // ---- boundary: the last LAGT tiles of the last block, the stores, the accumulators
        if (active) {
#pragma unroll
            for (int t = 0; t < LAGT; ++t) {
                const int j = TILES + t - LAGT, jn = j / TM, jm = j % TM;
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], s_old[jm], acc[jm][jn][q]);
            }
            uint16_t *C = p.out + (int64_t)T.g * p.c_gs;
            const int m_row = T.m0 + wm * (BM / Cfg::kWM) + li;
            const int n_base = T.n0 + wn * (BN / WN) + 8 * kg;
            const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int m = m_row + mt * 16;
                if (m >= T.M) continue;
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
                        if (p.out_nt == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(crow + n), "v"(pk) : "memory");
                        else if (p.out_nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(crow + n), "v"(pk) : "memory");
                        else if (p.out_nt == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(crow + n), "v"(pk) : "memory");
                        else *(v4i *)(crow + n) = pk;
                    } else {
                        const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (n + q < p.n) crow[n + q] = e[q];
                    }
                }
            }
        }
        if (!has_next) break;
        // ---- the next tile becomes the current one: its block 0 sits in stage `cur`, its block 1 is on its way into `nxt`
        const bool was_active = active;
        T = Tn; local = local_n;
        a_rsrc = a_rsrc_n; b_rsrc = b_rsrc_n;
        load_offsets(T);
        active = rows_present(T);
        clear_tile();
        // (a wave that multiplied the previous tile has this block's fragments already: converted in place during that tile's last
        //  k block; its s_cur is this block's; one that had no rows there sets them up now)
        if (active && !was_active) first_fragments(smem + cur * Cfg::STAGE_BYTES);
    }
    wait_vmcnt<0>();   // the refills past the last tile (zeros) land in LDS nobody reads: drain them before exit
}

}  // namespace dga
