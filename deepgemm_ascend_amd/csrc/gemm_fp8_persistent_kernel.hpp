// Persistent form of the loader-wave fp8 tile kernel (gemm_fp8_kernel.hpp, Cfg::kLC builds): one workgroup per CU walks
// a strided list of tiles, and the LDS ring does not stop at a tile boundary.
//
// Why.  With one workgroup per tile, every tile pays an un-overlapped start (kernel-argument reads, descriptor set-up, the
// first stages' HBM round trip with nothing in flight) and an un-overlapped end (the bf16 store burst while the CU's
// LDS-DMA queue is empty).  On the masked grouped weight stream that is 8 tiles per CU x ~5 us = ~5 % of the launch
// (profiles/r02_grouped_ablation.txt).  Here the four loader waves run STAGES-1 k blocks ahead of the computing waves
// over the FLATTENED (tile, k block) sequence: while the computing waves convert and store tile i, the loaders already
// have the first blocks of tile i+1 in flight, and the next tile's set-up (mask, row table, descriptors) is done by waves
// that have nothing else to do.  Both kinds of wave derive the same tile list from blockIdx and masked_m alone, so the
// one barrier per k block pairs up without any hand-shake beyond what the one-tile kernel has.  (Like every input,
// masked_m / m_indices must not be written while the launch runs: the two kinds of wave read them separately, and lists that
// disagree would leave a barrier unmatched.)
//
// Same arithmetic, same order, same bits as the one-tile builds (tests/test_grouped_gpu.py compares them byte for byte).
// Counterpart in the reference: its kernel is persistent by construction -- one block per AI core, which walks the
// m_parts x n_parts tiles of its section in the `mi` / `ni` loops with double-buffered L1 across them
// (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:160-198).
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

template <class Cfg, bool KTAIL>
__global__ void __launch_bounds__(Cfg::NT) gemm_fp8_blockscaled_nt_persistent_kernel(const GemmParams p)
{
    static_assert(Cfg::kLC && Cfg::STAGES >= 3, "loader waves, at least two refills in flight");
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN;
    constexpr int TM = Cfg::TM, TN = Cfg::TN;
    constexpr int STG = Cfg::STAGES, LOADS = Cfg::LOADS_PER_STAGE, DNT = Cfg::DNT;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= Cfg::kWM * WN;
    const int KB = p.kb_n;

    // ---- this workgroup's tile list: the raster is cut into one contiguous chunk per XCD (blocks b, b+8, ... share an XCD
    //      and its L2), and the workgroups of an XCD walk their chunk together, `step` tiles per round -- the order the
    //      hardware dispatcher gives the one-tile kernel
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
    struct Tile { int g, bg, M, m0, n0; };
    // masked_m / m_indices are read through the constant address space: they are inputs nobody writes during the launch,
    // and only so does the compiler use scalar loads (s_load, lgkmcnt) for them.  Behind the "memory" clobbers of the DMA
    // and wait statements it would otherwise fall back to a vector load + vmcnt(0) for a uniform address -- which in a
    // loader wave drains every refill in flight, and in a computing wave waits for the previous tile's stores.
    typedef const __attribute__((address_space(4))) int32_t *const_i32_ptr;
    const const_i32_ptr masked_m_c = (const_i32_ptr)p.masked_m;
    const const_i32_ptr m_indices_c = (const_i32_ptr)p.m_indices;
    // next tile at or behind position `local` of the chunk that has anything to do (wave-uniform: scalar loads only)
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
            int bg = g;
            if (p.m_indices) {                  // contiguous-grouped layout (tile height <= the segment alignment)
                bg = m_indices_c[m0];
                if (bg < 0 || bg >= p.b_groups) continue;
            }
            t.g = g; t.bg = bg; t.M = M; t.m0 = m0; t.n0 = tn * BN;
            return true;
        }
        return false;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    constexpr uint32_t kOutOfRange = 0x80000000u;

    if (loader) {
        // ================= loader waves: the LDS-DMA of every stage, STG-1 k blocks ahead of the barrier =================
        const int dwave = wave - Cfg::kWM * WN;
        const int dtid = dwave * 64 + lane;   // (not tid & (DNT-1): the loader waves need not start at a multiple of DNT threads)
        const int a_col = ((dtid & 7) ^ swz_a(dtid >> 3)) * 16;
        const int b_col = ((dtid & 7) ^ swz_b(dtid >> 3)) * 16;
        uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
        const float *sc_src[Cfg::SC_ITERS];
        v4i a_rsrc, b_rsrc;
        // Indexed form: the row-table entries a lane needs for a tile (one per A piece, one per scale piece) are fetched a
        // whole tile ahead, by LDS-DMA into this wave's own slots behind the stage ring -- a plain vector load here would
        // make the compiler wait vmcnt(0) in front of its first use, i.e. drain both refills in flight at every tile
        // boundary (measured: the indexed stream 6 % behind the packed one).  Only the low dword of an entry is read
        // (the launcher bounds rows * lda below 2^31).
        constexpr int IDX_PER_LANE = Cfg::A_ITERS + Cfg::SC_ITERS;
        const uint32_t idx_lds = lds0 + STG * Cfg::STAGE_BYTES + dwave * IDX_PER_LANE * 256;
        const uint32_t *idx_mine = (const uint32_t *)(smem + STG * Cfg::STAGE_BYTES + dwave * IDX_PER_LANE * 256) + lane;
        auto prefetch_idx = [&](const Tile &t) {
            const int64_t *ridx = p.row_index + (int64_t)t.g * p.m;
#pragma unroll
            for (int it = 0; it < Cfg::A_ITERS; ++it) {
                const int row = (it * DNT + dtid) >> 3;
                dma4(ridx + min(t.m0 + row, t.M - 1), idx_lds + it * 256);
            }
#pragma unroll
            for (int it = 0; it < Cfg::SC_ITERS; ++it)
                dma4(ridx + min(t.m0 + it * DNT + dtid, t.M - 1), idx_lds + (Cfg::A_ITERS + it) * 256);
        };
        // Weights that one CU reads once -- the weight stream of the grouped layouts -- go past the L2's retention
        // (non-temporal loads), which together with non-temporal output stores leaves the L2 to the A rows: see the
        // launcher (dga_launch.hip) for the measurements.  GemmParams::b_nt: 0 never (dense rasters share their B panels:
        // +10..25 %), 1 always, 2 per tile by its row count (<= 48 rows; what pays with default-policy stores).
        bool b_nt = p.b_nt == 1;
        auto setup = [&](const Tile &t) {
            if (p.b_nt == 2) b_nt = t.M <= 48;   // (between 48 and 112 rows the two policies are within a per cent of each other)
            const bool ridx = p.row_index != nullptr;
            const uint8_t *A = p.a + (int64_t)t.g * p.a_gs;
            const uint8_t *B = p.b + (int64_t)t.bg * p.b_gs;
            const float *SFA = p.sfa + (int64_t)t.g * p.sfa_gs;
            const float *SFB = p.sfb + (int64_t)t.bg * p.sfb_gs;
#pragma unroll
            for (int it = 0; it < Cfg::A_ITERS; ++it) {
                const int row = (it * DNT + dtid) >> 3;
                const int rr = min(row, t.M - 1 - t.m0);
                a_voff[it] = row < t.M - t.m0
                                 ? (ridx ? idx_mine[it * 64] : (uint32_t)rr) * (uint32_t)p.lda + a_col
                                 : kOutOfRange;   // rows at or beyond masked_m: nothing fetched, the LDS bytes are zero-filled
            }
#pragma unroll
            for (int it = 0; it < Cfg::B_ITERS; ++it) {
                const int row = (it * DNT + dtid) >> 3;
                b_voff[it] = (uint32_t)min(row, p.n - 1 - t.n0) * (uint32_t)p.ldb + b_col;
            }
            a_rsrc = ridx ? make_rsrc(A, p.a_bytes) : make_rsrc(A + (int64_t)t.m0 * p.lda, (int64_t)(t.M - t.m0) * p.lda);
            b_rsrc = make_rsrc(B + (int64_t)t.n0 * p.ldb, (int64_t)(p.n - t.n0) * p.ldb);
#pragma unroll
            for (int it = 0; it < Cfg::SC_ITERS; ++it) {
                const int s = it * DNT + dtid;
                if (s < BM) {
                    const int mr = min(t.m0 + s, t.M - 1);
                    sc_src[it] = SFA + (ridx ? (int64_t)idx_mine[(Cfg::A_ITERS + it) * 64] : (int64_t)mr) * p.sfa_ld;
                } else {
                    const int nb = min(t.n0 / 128 + min(s - BM, 7), p.nb_n - 1);
                    sc_src[it] = SFB + (int64_t)nb * p.kb_n;
                }
            }
        };
        // the whole refill of one stage: B and scale pieces first, then A (gemm_fp8_kernel.hpp, same order and count)
        auto issue_block = [&](int stage, int kb) {
            const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES + dwave * 1024;
            const int k0 = kb * 128;
#pragma unroll
            for (int it = 0; it < Cfg::B_ITERS; ++it) {
                uint32_t voff = b_voff[it];
                if constexpr (KTAIL) voff = (k0 + b_col < p.k) ? voff : kOutOfRange;
                if (b_nt) dma16_nt(voff, b_rsrc, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
                else dma16(voff, b_rsrc, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
            }
#pragma unroll
            for (int it = 0; it < Cfg::SC_ITERS; ++it)
                dma4(sc_src[it] + kb, lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES + (it * DNT + dwave * 64) * 4);
#pragma unroll
            for (int it = 0; it < Cfg::A_ITERS; ++it) {
                uint32_t voff = a_voff[it];
                if constexpr (KTAIL) voff = (k0 + a_col < p.k) ? voff : kOutOfRange;
                dma16(voff, a_rsrc, (uint32_t)k0, sa + it * DNT * 16);
            }
        };
        int local = slot;
        Tile t, tn;
        bool have = seek(local, t);
        if (!have) return;                       // the computing waves find the same empty list: no barrier is ever posted
        int local_n = local + step;
        bool have_n = seek(local_n, tn);         // the tile after: its row-table entries are fetched while this one streams
        if (p.row_index) {
            prefetch_idx(t);
            wait_vmcnt<0>();                     // first tile: nothing in flight yet, nothing to drain
        }
        setup(t);
        auto prefetch_next = [&]() {
            if (p.row_index && have_n) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // set-up has read its slots before they are overwritten
                prefetch_idx(tn);
            }
        };
        prefetch_next();
        int kb = 0, fill = 0, ahead = 0;         // ahead = blocks issued - barriers passed
        auto issue_next = [&]() {
            issue_block(fill, kb);
            fill = fill + 1 == STG ? 0 : fill + 1;
            ++ahead;
            if (++kb == KB) {                    // on to the next tile: its set-up is off the computing waves' path
                kb = 0;
                have = have_n;
                if (have) {
                    t = tn;
                    local_n += step;
                    have_n = seek(local_n, tn);
                    if (p.row_index) {
                        // this tile's entries were requested a tile ago, KB refills back: with KB >= STG-1 they are older
                        // than everything that may still fly (a wait that is already met in the steady state)
                        if (KB >= STG - 1) wait_vmcnt<(STG - 1) * LOADS>();
                        else wait_vmcnt<0>();
                    }
                    setup(t);
                    prefetch_next();
                }
            }
        };
        for (int d = 0; d < STG - 1 && have; ++d) issue_next();
        while (ahead > 0) {
            // block (issued - ahead) must have landed before its barrier; while the list lasts exactly STG-1 refills
            // are out, the youngest STG-2 may still fly.  Behind the last refill every wait is a full one.
            if (have) wait_vmcnt<(STG - 2) * LOADS>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            --ahead;
            if (have) issue_next();
        }
        return;
    }

    // ================= computing waves =================
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kg = lane >> 4;
    const int a_row = wm * (BM / Cfg::kWM) + li;
    const int a_off0 = a_row * 128 + ((kg ^ swz_a(a_row)) * 16);
    const int a_off1 = a_row * 128 + (((kg + 4) ^ swz_a(a_row)) * 16);
    const int b_row = wn * (BN / WN) + 8 * (li >> 2) + (li & 3);
    const int b_off0 = Cfg::A_BYTES + b_row * 128 + ((kg ^ swz_b(b_row)) * 16);
    const int b_off1 = Cfg::A_BYTES + b_row * 128 + (((kg + 4) ^ swz_b(b_row)) * 16);
    const int sa_off = Cfg::A_BYTES + Cfg::B_BYTES + (wm * (BM / Cfg::kWM) + li) * 4;
    const int sb_off = Cfg::A_BYTES + Cfg::B_BYTES + (BM + (wn * (BN / WN)) / 128) * 4;

    int local = slot, stage = 0;
    Tile t;
    while (seek(local, t)) {
        const int64_t *ridx = p.row_index ? p.row_index + (int64_t)t.g * p.m : nullptr;
        uint16_t *C = p.out + (int64_t)t.g * p.c_gs;
        const int m_row = t.m0 + wm * (BM / Cfg::kWM) + li;
        // indexed form: the destination rows are requested now and first looked at in the epilogue -- anything that
        // touched the loaded values here would put a vmcnt(0) (this load AND the previous tile's stores) in front of the
        // tile's first barrier, which the loader waves and every other computing wave would then wait for
        int64_t out_row[TM];
        if (ridx) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) out_row[mt] = ridx[min(m_row + mt * 16, t.M - 1)];
        }
        const bool wave_has_rows = t.m0 + wm * (BM / Cfg::kWM) < t.M;   // wave-uniform
        v4f acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

        for (int kb = 0; kb < KB; ++kb) {
            __builtin_amdgcn_s_barrier();            // block landed everywhere, the stage refilled next is free
            asm volatile("" ::: "memory");
            const uint8_t *st = smem + stage * Cfg::STAGE_BYTES;
            stage = stage + 1 == STG ? 0 : stage + 1;
            if (!wave_has_rows) continue;            // every row of this wave's m range is masked: it only keeps the barrier
            // the plain loop's k block (gemm_fp8_kernel.hpp, PP = 0): first MFMA waits only for its own operands, B fragments
            // rotate through two register sets, promotion FMAs run LAG steps behind the MFMAs
            constexpr int STEPS = TM * TN, LAG = 3, RING = LAG + 1;
            v4f part[RING];
            v8i bf[2];
            v8i af[TM];
            float s[TM];
            {
                const v4i lo = *(const v4i *)(st + b_off0);
                const v4i hi = *(const v4i *)(st + b_off1);
                bf[0] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const v4i lo = *(const v4i *)(st + a_off0 + mt * 2048);
                const v4i hi = *(const v4i *)(st + a_off1 + mt * 2048);
                af[mt] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (mt == 0) __builtin_amdgcn_sched_barrier(0);
            }
            const float sfb_v = *(const float *)(st + sb_off);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) s[mt] = *(const float *)(st + sa_off + mt * 64);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < STEPS + LAG; ++i) {
                if (i < STEPS) {
                    const int nt = i / TM, mt = i % TM;
                    part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        bf[nt & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (mt == 0 && nt + 1 < TN) {
                        const int boff = ((nt + 1) >> 1) * 4096 + ((nt + 1) & 1) * 512;
                        const v4i lo = *(const v4i *)(st + b_off0 + boff);
                        const v4i hi = *(const v4i *)(st + b_off1 + boff);
                        bf[(nt + 1) & 1] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
                }
                if (i == LAG) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) s[mt] *= sfb_v;  // two-level scale: sfa[m,kb] * sfb[n/128,kb]
                }
                if (i >= LAG) {
                    const int j = i - LAG, nt = j / TM, mt = j % TM;
                    const v4f pr = part[j % RING];
                    acc[mt][nt].x = __builtin_fmaf(pr.x, s[mt], acc[mt][nt].x);
                    acc[mt][nt].y = __builtin_fmaf(pr.y, s[mt], acc[mt][nt].y);
                    acc[mt][nt].z = __builtin_fmaf(pr.z, s[mt], acc[mt][nt].z);
                    acc[mt][nt].w = __builtin_fmaf(pr.w, s[mt], acc[mt][nt].w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue: lane owns row m, columns n0w + 32*j + 8*(lane>>4) + [0,8); the loaders are already filling the
        //      ring with the next tile's first blocks
        if (wave_has_rows) {
            const int n_base = t.n0 + wn * (BN / WN) + 8 * kg;
            const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
            // contiguous layout: the group index of this lane's rows, all requested before the first store goes out (a
            // load between the stores would wait for every store in front of it)
            int row_group[TM];
            if (p.m_indices) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) row_group[mt] = p.m_indices[min(m_row + mt * 16, t.M - 1)];
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int m = m_row + mt * 16;
                if (m >= t.M) continue;
                if (p.m_indices && row_group[mt] != t.bg) continue;  // a padding row: untouched
                uint16_t *crow = C + (ridx ? out_row[mt] : (int64_t)m) * p.ldc;
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
                        // the weight stream's outputs are written once and not read again here: stored non-temporally they do
                        // not take L2 lines from the A rows the expert's other tiles re-read (-3.5 % on 256 x (128, 7168, 2048))
                        // (inline asm: written as __builtin_nontemporal_store beside a plain store, the two branches are merged
                        //  by the compiler and the hint is lost)
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
        local += step;
    }
}

}  // namespace dga
