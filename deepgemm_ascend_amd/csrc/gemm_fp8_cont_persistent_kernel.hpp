// Persistent form of the continuous-pipeline 256x256 kernel (gemm_fp8_kernel.hpp, PP = 2): one workgroup per CU walks its
// share of a dense raster of FULL tiles, and the pipeline does not stop at a tile boundary.
//
// Why.  The one-tile kernel spends ~10 us per tile outside its main loop (first-stage latency with nothing in flight, the
// store burst with the LDS-DMA queue empty, the launch gap): 17 % of a 4096^3 launch, and again for every further tile a CU
// runs (8192^3: four per CU).  Here the refill slots of a tile's last two k blocks fetch the NEXT tile's first two blocks
// (its descriptors are set up a tile ahead), the in-place fragment reloads at the end of the last k block pick up the next
// tile's block 0 like any other block, and the tile boundary costs the drain of three promotions, the bf16 stores and the
// clearing of the accumulators -- while block 1 of the next tile is already landing.
//
// Restrictions (the launcher checks them, dga_launch_menu_d.hip): dense problem (one group, no mask, no index), M and N
// multiples of 256, K a multiple of 128.  Same arithmetic in the same order as the one-tile build: identical output bytes
// (tests/test_gemm_gpu.py).
// Counterpart in the reference: one block per AI core looping over its tiles with the L1 double buffer kept primed across
// them (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:160-198, 240-330).
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) gemm_fp8_blockscaled_nt_cont_persistent_kernel(const GemmParams p)
{
    static_assert(Cfg::kBM == 256 && Cfg::kBN == 256 && Cfg::kWM == 4 && Cfg::kWN == 2 && Cfg::STAGES == 2 && !Cfg::kLC,
                  "the continuous pipeline's tile");
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN;
    constexpr int NT = Cfg::NT, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int STEPS = TM * TN, LAG = 3, RING = 4;
    constexpr int SB = STEPS - TM - 1;          // barrier step (just before the in-place reloads)
    constexpr int NL = Cfg::LOADS_PER_STAGE;
    constexpr int TAIL_DMA = (TM < NL / 2 ? TM : NL / 2);                                            // as in the one-tile build
    constexpr int HEAD_STEPS = ((STEPS * 9) / 32 > NL - TAIL_DMA ? (STEPS * 9) / 32 : NL - TAIL_DMA);
    static_assert(STEPS % RING == 0 && STEPS > 2 * TM + LAG, "ring positions must line up across k blocks");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int KB = p.kb_n;

    // ---- tile list (as in gemm_fp8_persistent_kernel.hpp): the XCD's contiguous chunk of the raster, strided by its workgroups
    const int total = p.launch_tiles > 0 ? p.launch_tiles : p.tiles_m * p.tiles_n;
    int first = 0, count = total, step = gridDim.x, local = blockIdx.x;
    if (p.xcd_remap) {
        const int xcd = blockIdx.x & 7, q = total >> 3, r = total & 7;
        first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        count = q + (xcd < r ? 1 : 0);
        step = ((int)gridDim.x - xcd + 7) >> 3;
        local = blockIdx.x >> 3;
    }
    if (local >= count) return;
    auto tile_origin = [&](int t_in, int &m0, int &n0) {
        const int gm = p.raster_group;
        const int per = gm * p.tiles_n;
        const int band = t_in / per;
        const int row0 = band * gm;
        const int rows = min(p.tiles_m - row0, gm);
        const int loc = t_in - band * per;
        m0 = (row0 + loc % rows) * BM;
        n0 = (loc / rows) * BN;
    };

    // ---- per-lane DMA offsets: every tile is a full one, so they do not depend on the tile
    constexpr int DNT = Cfg::DNT;
    const int dtid = tid & (DNT - 1);
    const int a_col = ((dtid & 7) ^ swz_a(dtid >> 3)) * 16;
    const int b_col = ((dtid & 7) ^ swz_b(dtid >> 3)) * 16;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::A_ITERS; ++it) a_voff[it] = (uint32_t)((it * DNT + dtid) >> 3) * (uint32_t)p.lda + a_col;
#pragma unroll
    for (int it = 0; it < Cfg::B_ITERS; ++it) b_voff[it] = (uint32_t)((it * DNT + dtid) >> 3) * (uint32_t)p.ldb + b_col;
    static_assert(Cfg::SC_ITERS == 1, "one scale piece per stage");
    // scale slots: [0,BM) = sfa rows of the tile, [BM, BM+8) = its sfb blocks, rest = padding (re-reads the last sfb block)
    auto scale_src = [&](int m0, int n0) -> const float * {
        const int s = dtid;
        if (s < BM) return p.sfa + (int64_t)(m0 + s) * p.sfa_ld;
        const int nb = min(n0 / 128 + min(s - BM, 7), p.nb_n - 1);
        return p.sfb + (int64_t)nb * p.kb_n;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // descriptors of the tile being computed and of the one after it
    int m0, n0, m0n, n0n;
    tile_origin(first + local, m0, n0);
    bool have_next = local + step < count;
    if (have_next) tile_origin(first + local + step, m0n, n0n);
    else { m0n = m0; n0n = n0; }     // no next tile: the run-ahead refills re-read this one's bytes into stages nobody consumes
    v4i a_rsrc = make_rsrc(p.a + (int64_t)m0 * p.lda, (int64_t)(p.m - m0) * p.lda);
    v4i b_rsrc = make_rsrc(p.b + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
    v4i a_rsrc_n = make_rsrc(p.a + (int64_t)m0n * p.lda, (int64_t)(p.m - m0n) * p.lda);
    v4i b_rsrc_n = make_rsrc(p.b + (int64_t)n0n * p.ldb, (int64_t)(p.n - n0n) * p.ldb);
    const float *sc_src = scale_src(m0, n0), *sc_src_n = scale_src(m0n, n0n);

    // one LDS-DMA wave-instruction of a stage; `ra` / `rb` / `sc` / `kb` name the tile and k block it belongs to
    auto issue_one = [&](int idx, int stage, const v4i &ra, const v4i &rb, const float *sc, int kb) {
        const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES + wave * 1024;
        const int k0 = kb * 128;
        if (idx < Cfg::A_ITERS) {
            dma16(a_voff[idx], ra, (uint32_t)k0, sa + idx * DNT * 16);
        } else if (idx < Cfg::A_ITERS + Cfg::B_ITERS) {
            const int it = idx - Cfg::A_ITERS;
            dma16(b_voff[it], rb, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
        } else {
            dma4(sc + min(kb, KB - 1), lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES + wave * 64 * 4);
        }
    };

    // ---- per-lane fragment read offsets (bytes inside a stage)
    const int li = lane & 15, kg = lane >> 4;
    const int a_row = wm * (BM / Cfg::kWM) + li;
    const int a_off0 = a_row * 128 + ((kg ^ swz_a(a_row)) * 16);
    const int a_off1 = a_row * 128 + (((kg + 4) ^ swz_a(a_row)) * 16);
    const int b_row = wn * (BN / WN) + 8 * (li >> 2) + (li & 3);
    const int b_off0 = Cfg::A_BYTES + b_row * 128 + ((kg ^ swz_b(b_row)) * 16);
    const int b_off1 = Cfg::A_BYTES + b_row * 128 + (((kg + 4) ^ swz_b(b_row)) * 16);
    const int sa_off = Cfg::A_BYTES + Cfg::B_BYTES + (wm * (BM / Cfg::kWM) + li) * 4;
    const int sb_off = Cfg::A_BYTES + Cfg::B_BYTES + (BM + (wn * (BN / WN)) / 128) * 4;
    auto read_b = [&](const uint8_t *st, int nt) {
        const int boff = (nt >> 1) * 4096 + (nt & 1) * 512;
        const v4i lo = *(const v4i *)(st + b_off0 + boff);
        const v4i hi = *(const v4i *)(st + b_off1 + boff);
        return v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    };
    auto read_a = [&](const uint8_t *st, int mt) {
        const v4i lo = *(const v4i *)(st + a_off0 + mt * 2048);
        const v4i hi = *(const v4i *)(st + a_off1 + mt * 2048);
        return v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
    v4f part[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
    v8i bf[2], af[TM];
    float s[TM], s_prev[TM], s_next[TM], sfb_next = 0.f;

    // ---- prologue, once per workgroup: all of block 0, then the tail part of block 1 (of the first tile; a one-block tile's
    //      "block 1" is the next tile's block 0)
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) issue_one(idx, 0, a_rsrc, b_rsrc, sc_src, 0);
    {
        const bool nx = 1 >= KB;
#pragma unroll
        for (int idx = 0; idx < TAIL_DMA; ++idx)
            issue_one(idx, 1, nx ? a_rsrc_n : a_rsrc, nx ? b_rsrc_n : b_rsrc, nx ? sc_src_n : sc_src, nx ? 1 - KB : 1);
    }
    wait_vmcnt<TAIL_DMA>();
    barrier();
    bf[0] = read_b(smem, 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) af[mt] = read_a(smem, mt);
    {
        const float sfb0 = *(const float *)(smem + sb_off);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            s[mt] = *(const float *)(smem + sa_off + mt * 64) * sfb0;
            s_prev[mt] = 0.f;
            s_next[mt] = 0.f;
        }
    }

    int gb = 0;   // k blocks done by this workgroup: the stage of a block is its parity
    for (;;) {
        for (int kb = 0; kb < KB; ++kb, ++gb) {
            const uint8_t *st = smem + (gb & 1) * Cfg::STAGE_BYTES;
            const uint8_t *sn = smem + ((gb & 1) ^ 1) * Cfg::STAGE_BYTES;
            // whose blocks the refill slots of this k block fetch: block kb+1 (head part) and kb+2 (tail part) of this tile,
            // or blocks 0 / 1 of the next one
            const bool hn = kb + 1 >= KB, tn = kb + 2 >= KB;
            const v4i ha = hn ? a_rsrc_n : a_rsrc, hb = hn ? b_rsrc_n : b_rsrc;
            const v4i ta = tn ? a_rsrc_n : a_rsrc, tb = tn ? b_rsrc_n : b_rsrc;
            const float *hs = hn ? sc_src_n : sc_src, *ts = tn ? sc_src_n : sc_src;
            const int hk = hn ? kb + 1 - KB : kb + 1, tk = tn ? kb + 2 - KB : kb + 2;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                const int nt = i / TM, mt = i % TM;
                if (i == SB) {
                    wait_vmcnt<0>();
                    barrier();
                }
                part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    bf[nt & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < HEAD_STEPS) {
#pragma unroll
                    for (int j = (i * (NL - TAIL_DMA)) / HEAD_STEPS; j < ((i + 1) * (NL - TAIL_DMA)) / HEAD_STEPS; ++j)
                        issue_one(TAIL_DMA + j, (gb & 1) ^ 1, ha, hb, hs, hk);
                }
                if (i > SB) {
#pragma unroll
                    for (int j = ((i - SB - 1) * TAIL_DMA) / TM; j < ((i - SB) * TAIL_DMA) / TM; ++j)
                        issue_one(j, gb & 1, ta, tb, ts, tk);
                }
                if (mt == 0) {
                    if (nt + 1 < TN) bf[(nt + 1) & 1] = read_b(st, nt + 1);
                    else bf[(nt + 1) & 1] = read_b(sn, 0);
                }
                if (nt == TN - 1) {
                    af[mt] = read_a(sn, mt);
                    if (mt == 0) sfb_next = *(const float *)(sn + sb_off);
                    s_next[mt] = *(const float *)(sn + sa_off + mt * 64);
                }
                if (i >= LAG) {
                    const int j = i - LAG, jn = j / TM, jm = j % TM;
                    const v4f pr = part[j % RING];
                    acc[jm][jn].x = __builtin_fmaf(pr.x, s[jm], acc[jm][jn].x);
                    acc[jm][jn].y = __builtin_fmaf(pr.y, s[jm], acc[jm][jn].y);
                    acc[jm][jn].z = __builtin_fmaf(pr.z, s[jm], acc[jm][jn].z);
                    acc[jm][jn].w = __builtin_fmaf(pr.w, s[jm], acc[jm][jn].w);
                } else {
                    const int j = STEPS - LAG + i, jn = j / TM, jm = j % TM;  // previous k block's last steps
                    const v4f pr = part[j % RING];
                    acc[jm][jn].x = __builtin_fmaf(pr.x, s_prev[jm], acc[jm][jn].x);
                    acc[jm][jn].y = __builtin_fmaf(pr.y, s_prev[jm], acc[jm][jn].y);
                    acc[jm][jn].z = __builtin_fmaf(pr.z, s_prev[jm], acc[jm][jn].z);
                    acc[jm][jn].w = __builtin_fmaf(pr.w, s_prev[jm], acc[jm][jn].w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                s_prev[mt] = s[mt];
                s[mt] = s_next[mt] * sfb_next;
            }
        }
        // ---- tile boundary: the last LAG results of the last k block, then the stores.  Block 0 of the next tile sits in
        //      stage gb & 1 (its fragments and scales are already in registers), the tail part of its block 1 is in flight.
#pragma unroll
        for (int i = 0; i < LAG; ++i) {
            const int j = STEPS - LAG + i, jn = j / TM, jm = j % TM;
            const v4f pr = part[j % RING];
            acc[jm][jn].x = __builtin_fmaf(pr.x, s_prev[jm], acc[jm][jn].x);
            acc[jm][jn].y = __builtin_fmaf(pr.y, s_prev[jm], acc[jm][jn].y);
            acc[jm][jn].z = __builtin_fmaf(pr.z, s_prev[jm], acc[jm][jn].z);
            acc[jm][jn].w = __builtin_fmaf(pr.w, s_prev[jm], acc[jm][jn].w);
        }
        {
            const int m_row = m0 + wm * (BM / Cfg::kWM) + li;
            const int n_base = n0 + wn * (BN / WN) + 8 * kg;
            const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)p.out) & 15) == 0);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                uint16_t *crow = p.out + (int64_t)(m_row + mt * 16) * p.ldc;
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
                    if (vec_ok) {
                        if (p.out_nt == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(crow + n), "v"(pk) : "memory");
                        else if (p.out_nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(crow + n), "v"(pk) : "memory");
                        else if (p.out_nt == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(crow + n), "v"(pk) : "memory");
                        else *(v4i *)(crow + n) = pk;
                    } else {
                        const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                        for (int q = 0; q < 8; ++q) crow[n + q] = e[q];
                    }
                }
            }
        }
        if (!have_next) break;
        // ---- on to the next tile: its descriptors become the current ones, the tile after it is set up
        local += step;
        m0 = m0n; n0 = n0n;
        a_rsrc = a_rsrc_n; b_rsrc = b_rsrc_n; sc_src = sc_src_n;
        have_next = local + step < count;
        if (have_next) {
            tile_origin(first + local + step, m0n, n0n);
            a_rsrc_n = make_rsrc(p.a + (int64_t)m0n * p.lda, (int64_t)(p.m - m0n) * p.lda);
            b_rsrc_n = make_rsrc(p.b + (int64_t)n0n * p.ldb, (int64_t)(p.n - n0n) * p.ldb);
            sc_src_n = scale_src(m0n, n0n);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};   // (a NaN left here would meet s_prev = 0 below)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) s_prev[mt] = 0.f;
    }
    wait_vmcnt<0>();   // the run-ahead refills behind the last tile land in LDS nobody reads: drain them before exit
}

}  // namespace dga
