// Stream-K form of the continuous-pipeline 256x256 kernel: ONE launch, one workgroup per CU.  Whole rounds of the raster run as in
// gemm_fp8_cont_persistent_kernel.hpp (a workgroup walks its tiles, the LDS ring running straight through); the LAST, partial round
// -- R tiles for P workgroups -- is split along K into s = 2^floor(log2(P / R)) parts per tile, so that it costs a 1/s of a round
// instead of a whole one:
//   * part j of every remainder tile goes to the SAME group of workgroups of an XCD (logical index q -> part q / R, tile q % R):
//     the workgroups of an XCD then read the same k slice of the operand panels they share at the same time, as in a whole round.
//     (A first form cut the raster's k blocks into P contiguous runs: every workgroup at its own k offset -- the L2 stopped serving
//     them, 2.3 us per k block instead of 1.25: profiles/r05_streamk_skewed_ab.txt.)
//   * every part leaves its fp32 partial tile in the workspace (256 KB per workgroup, one slot each), raises its flag, waits for the
//     flags of the tile's other parts, and then reduces ITS 1/s of the tile's accumulator registers over the s partials, in k order
//     (a fixed order: the result does not depend on timing), and stores those bf16 rows: per workgroup 256 KB written and 256 KB
//     read whatever s is.
// So a raster of 1.125 rounds costs 1.125 rounds of k blocks on every CU instead of two rounds on some.  Counterpart in the
// reference: kernel type 4, PaddingStreamkMatmulKernel -- Stream-K split of the k loop over all cores + StreamkReduceAdd over fp32
// partials (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_streamk_matmul_kernel.h:94-98; selection rule
// op_host/op_tiling/select_kernel.cpp:303-331: more blocks than cores, a remainder below 0.8 of the cores, k > 3072).
//
// A flag is raised before its workgroup waits for anything, but the wait itself holds a CU: every workgroup must be RESIDENT at
// once (one per CU; where a CU mask narrows the queue the launcher does not launch this kernel: dga_launch_menu_m.hip).
// A flag is "raised" when it holds this launch's 64-bit epoch value, so nothing has to be zeroed in front of an ordinary launch; a
// launch that is being captured into a graph (replays repeat the epoch) has its flags zeroed by a memset node instead (launcher).
// Restrictions (launcher): dense, M and N multiples of 256, K of 128, at least 2 k blocks per part.
// MATH = 0: the promotion form; MATH = 2: block scales in the MFMA's E8M0 operands (power-of-two scales), accumulate in place.
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

struct StreamKArgs {
    float *partials;               // [gridDim.x][256 * 256] fp32, one slot per workgroup
    unsigned long long *flags;     // [gridDim.x]: `epoch` = "this launch's partial is in the slot"
    unsigned long long epoch;      // a value no earlier launch used (and the memory is unlikely to hold: 64 mixed bits); graph replays
                                   // repeat it, which is why a consumer puts the flag back to 0
};

template <class Cfg, int MATH>
__global__ void __launch_bounds__(Cfg::NT) gemm_fp8_blockscaled_nt_streamk_kernel(const GemmParams p, const StreamKArgs sk)
{
    static_assert(Cfg::kBM == 256 && Cfg::kBN == 256 && Cfg::kWM == 4 && Cfg::kWN == 2 && Cfg::STAGES == 2 && !Cfg::kLC,
                  "the continuous pipeline's tile");
    static_assert(MATH == 0 || MATH == 2, "promotion form or hardware-scale form");
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN;
    constexpr int NT = Cfg::NT, TM = Cfg::TM, TN = Cfg::TN;
    constexpr int STEPS = TM * TN, LAG = 3, RING = 4;
    constexpr int SB = STEPS - TM - 1;
    constexpr int NL = Cfg::LOADS_PER_STAGE;
    constexpr int TAIL_DMA = (TM < NL / 2 ? TM : NL / 2);
    constexpr int HEAD_STEPS = ((STEPS * 9) / 32 > NL - TAIL_DMA ? (STEPS * 9) / 32 : NL - TAIL_DMA);
    constexpr int MINSEG = 2;
    constexpr int SLOT = BM * BN;      // floats per partial tile
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int KB = p.kb_n;

    // ---- this workgroup's segments: its whole tiles (the XCD's contiguous chunk of the raster's whole rounds, strided by the XCD's
    //      workgroups, as in the persistent kernel), then at most one part of a tile of the last, partial round
    const int P = gridDim.x;
    const int tiles = p.tiles_m * p.tiles_n;
    const int n_dp = __builtin_amdgcn_readfirstlane(tiles / P);   // whole tiles of this workgroup
    const int full = n_dp * P;                     // tiles of the whole rounds
    const int R = tiles - full;                    // tiles of the partial round (< P)
    int first = 0, step = P, local = blockIdx.x, q = blockIdx.x;
    if (p.xcd_remap && (P & 7) == 0) {
        const int xcd = blockIdx.x & 7;
        first = xcd * (full >> 3);
        step = P >> 3;
        local = blockIdx.x >> 3;
        q = xcd * (P >> 3) + local;                // logical index: the workgroups of an XCD are neighbours
    }
    // the split of the partial round: s parts per tile (a power of two, at most 16, at least 2 k blocks each); 1 = no split (the
    // first R workgroups take a whole tile each)
    int lg = 0;                                    // log2 of the parts per tile
    while (R > 0 && (2 << lg) * R <= P && (2 << lg) <= 16 && (KB >> (lg + 1)) >= MINSEG) ++lg;
    const int sp = 1 << lg;
    // (integer divisions run on the vector pipe: their results are uniform but sit in vector registers -- read them back, the DMA
    //  statements below want scalars)
    // the R * sp parts are spread evenly over the XCDs (R * sp / 8 each, rounded up): when they are fewer than the workgroups, the
    // idle ones are a few CUs of every XCD, not whole XCDs (160 whole tiles on five of the eight XCDs ran 27 % slower than on eight)
    int qt = q;                                    // index of this workgroup's part in (part-major, tile-minor) order; >= R * sp: none
    if (p.xcd_remap && (P & 7) == 0) {
        const int per_xcd = (R * sp + 7) >> 3;
        qt = local < per_xcd ? (int)(blockIdx.x & 7) * per_xcd + local : P;
    }
    const int kpart = R > 0 ? __builtin_amdgcn_readfirstlane(qt / R) : 0;
    const int rt = R > 0 ? __builtin_amdgcn_readfirstlane(qt % R) : 0;
    const bool has_tail = R > 0 && kpart < sp;
    const int tail_b = has_tail ? (kpart * KB) >> lg : 0, tail_e = has_tail ? ((kpart + 1) * KB) >> lg : 0;
    const int n_seg = n_dp + (has_tail ? 1 : 0);
    if (n_seg == 0) return;
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
    // segment i: tile and k range
    auto seg_tile = [&](int i) { return i < n_dp ? first + local + i * step : full + rt; };

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
    auto scale_src = [&](int m0, int n0) -> const float * {
        const int s = dtid;
        if (s < BM) return p.sfa + (int64_t)(m0 + s) * p.sfa_ld;
        const int nb = min(n0 / 128 + min(s - BM, 7), p.nb_n - 1);
        return p.sfb + (int64_t)nb * p.kb_n;
    };
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // current segment: tile t, k blocks [kb_b, kb_e); the next one is set up a segment ahead
    int seg = 0;
    int t = seg_tile(0);
    int kb_b = n_dp > 0 ? 0 : tail_b, kb_e = n_dp > 0 ? KB : tail_e;
    int kb_b_n = 0;                               // first k block of the next segment
    int m0, n0, m0n, n0n;
    tile_origin(t, m0, n0);
    bool have_next = n_seg > 1;
    if (have_next) { tile_origin(seg_tile(1), m0n, n0n); kb_b_n = 1 < n_dp ? 0 : tail_b; }
    else { m0n = m0; n0n = n0; }
    v4i a_rsrc = make_rsrc(p.a + (int64_t)m0 * p.lda, (int64_t)(p.m - m0) * p.lda);
    v4i b_rsrc = make_rsrc(p.b + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
    v4i a_rsrc_n = make_rsrc(p.a + (int64_t)m0n * p.lda, (int64_t)(p.m - m0n) * p.lda);
    v4i b_rsrc_n = make_rsrc(p.b + (int64_t)n0n * p.ldb, (int64_t)(p.n - n0n) * p.ldb);
    const float *sc_src = scale_src(m0, n0), *sc_src_n = scale_src(m0n, n0n);

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
    auto e8m0 = [](float v) { return (int)((uint32_t)__builtin_bit_cast(int, v) >> 23); };

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
    int sae[TM], sbe = 0;

    // ---- prologue: all of the first segment's first block, then the tail part of its second (a segment has >= 2 blocks)
#pragma unroll
    for (int idx = 0; idx < NL; ++idx) issue_one(idx, 0, a_rsrc, b_rsrc, sc_src, kb_b);
#pragma unroll
    for (int idx = 0; idx < TAIL_DMA; ++idx) issue_one(idx, 1, a_rsrc, b_rsrc, sc_src, kb_b + 1);
    wait_vmcnt<TAIL_DMA>();
    barrier();
    bf[0] = read_b(smem, 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) af[mt] = read_a(smem, mt);
    {
        const float sfb0 = *(const float *)(smem + sb_off);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const float sa0 = *(const float *)(smem + sa_off + mt * 64);
            s[mt] = sa0 * sfb0;
            sae[mt] = e8m0(sa0);
            s_prev[mt] = 0.f;
            s_next[mt] = 0.f;
        }
        sbe = e8m0(sfb0);
    }

    // bf16 rows of one whole tile
    auto store_tile = [&](v4f (&c)[TM][TN], int tm0, int tn0) {
        const int m_row = tm0 + wm * (BM / Cfg::kWM) + li;
        const int n_base = tn0 + wn * (BN / WN) + 8 * kg;
        const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)p.out) & 15) == 0);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            uint16_t *crow = p.out + (int64_t)(m_row + mt * 16) * p.ldc;
#pragma unroll
            for (int j = 0; j < TN / 2; ++j) {
                const int n = n_base + 32 * j;
                const v4f lo = c[mt][2 * j], hi = c[mt][2 * j + 1];
                const v2bf h0 = __builtin_convertvector(v2f{lo.x, lo.y}, v2bf);
                const v2bf h1 = __builtin_convertvector(v2f{lo.z, lo.w}, v2bf);
                const v2bf h2 = __builtin_convertvector(v2f{hi.x, hi.y}, v2bf);
                const v2bf h3 = __builtin_convertvector(v2f{hi.z, hi.w}, v2bf);
                const v4i pk = v4i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1),
                                   __builtin_bit_cast(int, h2), __builtin_bit_cast(int, h3)};
                if (vec_ok) {
                    *(v4i *)(crow + n) = pk;
                } else {
                    const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                    for (int q = 0; q < 8; ++q) crow[n + q] = e[q];
                }
            }
        }
    };

    int gb = 0;   // k blocks done by this workgroup: the stage of a block is its parity
    for (;;) {
        for (int kb = kb_b; kb < kb_e; ++kb, ++gb) {
            const int stg = __builtin_amdgcn_readfirstlane(gb & 1);   // (uniform; said so, or the DMA's LDS address is built on the vector pipe)
            const uint8_t *st = smem + stg * Cfg::STAGE_BYTES;
            const uint8_t *sn = smem + (stg ^ 1) * Cfg::STAGE_BYTES;
            // whose blocks the refill slots of this k block fetch: blocks kb+1 (head part) and kb+2 (tail part) of this segment, or
            // blocks 0 / 1 of the next one (which starts its tile)
            const bool hn = kb + 1 >= kb_e, tn = kb + 2 >= kb_e;
            const v4i ha = hn ? a_rsrc_n : a_rsrc, hb = hn ? b_rsrc_n : b_rsrc;
            const v4i ta = tn ? a_rsrc_n : a_rsrc, tb = tn ? b_rsrc_n : b_rsrc;
            const float *hs = hn ? sc_src_n : sc_src, *ts = tn ? sc_src_n : sc_src;
            const int hk = hn ? kb_b_n + kb + 1 - kb_e : kb + 1, tk = tn ? kb_b_n + kb + 2 - kb_e : kb + 2;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                const int nt = i / TM, mt = i % TM;
                if (i == SB) {
                    wait_vmcnt<0>();
                    barrier();
                }
                if constexpr (MATH == 2)
                    acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[nt & 1], af[mt], acc[mt][nt], 0, 0, 0, sbe, 0, sae[mt]);
                else
                    part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        bf[nt & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < HEAD_STEPS) {
#pragma unroll
                    for (int j = (i * (NL - TAIL_DMA)) / HEAD_STEPS; j < ((i + 1) * (NL - TAIL_DMA)) / HEAD_STEPS; ++j)
                        issue_one(TAIL_DMA + j, stg ^ 1, ha, hb, hs, hk);
                }
                if (i > SB) {
#pragma unroll
                    for (int j = ((i - SB - 1) * TAIL_DMA) / TM; j < ((i - SB) * TAIL_DMA) / TM; ++j)
                        issue_one(j, stg, ta, tb, ts, tk);
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
                if constexpr (MATH == 0) {
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
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                if constexpr (MATH == 2) {
                    sae[mt] = e8m0(s_next[mt]);
                } else {
                    s_prev[mt] = s[mt];
                    s[mt] = s_next[mt] * sfb_next;
                }
            }
            if constexpr (MATH == 2) sbe = e8m0(sfb_next);
        }
        // ---- segment boundary: the last LAG results of its last k block
        if constexpr (MATH == 0) {
#pragma unroll
            for (int i = 0; i < LAG; ++i) {
                const int j = STEPS - LAG + i, jn = j / TM, jm = j % TM;
                const v4f pr = part[j % RING];
                acc[jm][jn].x = __builtin_fmaf(pr.x, s_prev[jm], acc[jm][jn].x);
                acc[jm][jn].y = __builtin_fmaf(pr.y, s_prev[jm], acc[jm][jn].y);
                acc[jm][jn].z = __builtin_fmaf(pr.z, s_prev[jm], acc[jm][jn].z);
                acc[jm][jn].w = __builtin_fmaf(pr.w, s_prev[jm], acc[jm][jn].w);
            }
        }
        if (seg >= n_dp) break;     // the part of a partial-round tile is the LAST segment: finished below, outside the loop
        store_tile(acc, m0, n0);
        if (!have_next) break;
        // ---- on to the next segment: its descriptors become the current ones, the one after it is set up
        seg += 1;
        t = seg_tile(seg);
        kb_b = kb_b_n;
        kb_e = seg < n_dp ? KB : tail_e;
        m0 = m0n; n0 = n0n;
        a_rsrc = a_rsrc_n; b_rsrc = b_rsrc_n; sc_src = sc_src_n;
        have_next = seg + 1 < n_seg;
        if (have_next) {
            tile_origin(seg_tile(seg + 1), m0n, n0n);
            kb_b_n = seg + 1 < n_dp ? 0 : tail_b;
            a_rsrc_n = make_rsrc(p.a + (int64_t)m0n * p.lda, (int64_t)(p.m - m0n) * p.lda);
            b_rsrc_n = make_rsrc(p.b + (int64_t)n0n * p.ldb, (int64_t)(p.n - n0n) * p.ldb);
            sc_src_n = scale_src(m0n, n0n);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) s_prev[mt] = 0.f;
    }
    wait_vmcnt<0>();   // the run-ahead refills behind the last segment land in LDS nobody reads: drain them before exit
    if (has_tail) {
        if (sp == 1) {          // the partial round is not split: a whole tile
            store_tile(acc, m0, n0);
            return;
        }
        // ---- part `kpart` of remainder tile `rt`: the fp32 partial goes to this workgroup's slot, lane-linear (16 bytes per lane per
        //      accumulator tile), then the flag.  Every access to a slot or a flag carries sc1: it is served at the device-coherent
        //      level (a partial is read by workgroups of other XCDs), so neither side needs an L2 write-back or invalidate -- an
        //      acquire loop that invalidates the XCD's L2 on every poll took the operand panels of its neighbours with it.
        {
            float *slot = sk.partials + (int64_t)(kpart * R + rt) * SLOT + tid * 4;     // (one running pointer: an asm operand takes no immediate offset)
#ifndef DGA_SK_ABLATE_STORES      // (diagnostic builds only -- results are then wrong: the exchange with one cost removed, scripts/ubench)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slot), "v"(acc[mt][nt]) : "memory");
                    slot += NT * 4;
                    asm volatile("" : "+v"(slot));
                }
#endif
            wait_vmcnt<0>();                                        // this wave's rows have reached the coherent level ...
            barrier();                                              // ... every wave's have
            if (tid == 0) __hip_atomic_store(sk.flags + kpart * R + rt, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- this part's share of the reduction: accumulator tiles [kpart * 32 / sp, +32 / sp) of every lane (pairs of n-tiles: one
        //      16-byte bf16 store each), summed over the sp partials in k order
        const int per = (TM * TN) / sp;                  // 16, 8, 4 or 2
        for (int j = 0; j < sp; ++j)
                while (__hip_atomic_load(sk.flags + j * R + rt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(8);
        const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)p.out) & 15) == 0);
        for (int pi = 0; pi < per; pi += 2) {
            const int idx = kpart * per + pi, mt = idx / TN, nt = idx % TN;      // (nt is even)
            v4f lo = v4f{0.f, 0.f, 0.f, 0.f}, hi = v4f{0.f, 0.f, 0.f, 0.f};
            for (int j0 = 0; j0 < sp; j0 += 4) {           // four partials per round trip
                v4f ql[4], qh[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = min(j0 + u, sp - 1);
                    const float *src = sk.partials + (int64_t)(j * R + rt) * SLOT + (idx * NT + tid) * 4;
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(ql[u]) : "v"(src) : "memory");
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(qh[u]) : "v"(src + NT * 4) : "memory");
                }
                wait_vmcnt<0>();
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    asm volatile("" : "+v"(ql[u]), "+v"(qh[u]));     // (the loads' results are valid from here on)
                    if (j0 + u < sp) {
                        lo.x += ql[u].x; lo.y += ql[u].y; lo.z += ql[u].z; lo.w += ql[u].w;
                        hi.x += qh[u].x; hi.y += qh[u].y; hi.z += qh[u].z; hi.w += qh[u].w;
                    }
                }
            }
            const int m = m0 + wm * (BM / Cfg::kWM) + li + mt * 16;
            const int n = n0 + wn * (BN / WN) + 8 * kg + 32 * (nt >> 1);
            uint16_t *crow = p.out + (int64_t)m * p.ldc;
            const v2bf h0 = __builtin_convertvector(v2f{lo.x, lo.y}, v2bf);
            const v2bf h1 = __builtin_convertvector(v2f{lo.z, lo.w}, v2bf);
            const v2bf h2 = __builtin_convertvector(v2f{hi.x, hi.y}, v2bf);
            const v2bf h3 = __builtin_convertvector(v2f{hi.z, hi.w}, v2bf);
            const v4i pk = v4i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1), __builtin_bit_cast(int, h2), __builtin_bit_cast(int, h3)};
            if (vec_ok) {
                *(v4i *)(crow + n) = pk;
            } else {
                const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                for (int qq = 0; qq < 8; ++qq) crow[n + qq] = e[qq];
            }
        }
    }
}

}  // namespace dga
