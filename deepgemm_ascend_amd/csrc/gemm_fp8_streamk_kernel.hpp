// Stream-K form of the continuous-pipeline 256x256 kernel: ONE launch, one workgroup per CU, and the unit of work is the k block,
// not the tile.  The raster's tiles x k blocks are laid end to end and cut into gridDim.x equal runs; a workgroup walks its run
// with the LDS ring running straight through (as gemm_fp8_cont_persistent_kernel.hpp does across whole tiles).  A run may start in
// the middle of a tile and may end in the middle of another:
//   * a segment that does NOT start at k block 0 leaves its fp32 partial tile in the workspace (256 KB per workgroup, one slot
//     each) and raises the workgroup's flag -- it is the FIRST thing its workgroup does, so the partial is there long before it is
//     wanted;
//   * a segment that starts at k block 0 but stops short of the tile's end is the LAST thing its workgroup does: it then adds the
//     partials of the workgroups that hold the rest of the tile, in k order (a fixed order: the result does not depend on timing),
//     and stores the bf16 rows;
//   * whole tiles in between are stored as they are.
// So a raster of 1.125 rounds of tiles costs 1.125 rounds of k blocks on every CU instead of two rounds on some, and a raster of
// 0.6 rounds keeps every CU busy.  Counterpart in the reference: kernel type 4, PaddingStreamkMatmulKernel -- Stream-K split of the
// k loop over all cores + StreamkReduceAdd over fp32 partials
// (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_streamk_matmul_kernel.h:94-98; selection rule
// op_host/op_tiling/select_kernel.cpp:303-331).
//
// No workgroup ever waits for a workgroup that waits: a flag is raised by a FIRST segment, which waits for nothing.  (So the launch
// makes progress whatever subset of its workgroups is resident.)  A flag is "raised" when it holds this launch's 64-bit epoch value:
// nothing has to be zeroed in front of the launch (a memset costs a 5 us launch of its own), whatever the workspace held.
// Cuts are snapped so that no segment is shorter than two k blocks (the refill slots look two blocks ahead).
// Restrictions (launcher): dense, M and N multiples of 256, K of 128, at least 4 k blocks.
// MATH = 0: the promotion form; MATH = 2: block scales in the MFMA's E8M0 operands (power-of-two scales), accumulate in place.
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

struct StreamKArgs {
    float *partials;               // [gridDim.x][256 * 256] fp32, one slot per workgroup
    unsigned long long *flags;     // [gridDim.x]: `epoch` = "this launch's partial is in the slot"
    int debug;                     // diagnostics ($DGA_SK_DEBUG; results are then wrong): 1 no partial stores, 2 no partial loads, 4 no flag wait
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

    // ---- this workgroup's run of (tile, k block) units.  Logical index: workgroups of one XCD (blockIdx & 7) take neighbouring
    //      runs, so that the tiles an XCD works on at one time share operand panels in its L2
    const int P = gridDim.x;
    const int w = (p.xcd_remap && (P & 7) == 0) ? (int)(blockIdx.x & 7) * (P >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const uint32_t U = (uint32_t)p.tiles_m * (uint32_t)p.tiles_n * (uint32_t)KB;     // (the launcher keeps U * P below 2^31)
    auto cut = [&](int i) -> int {
        const uint32_t c = (uint32_t)i * U / (uint32_t)P;
        const uint32_t r = c % (uint32_t)KB;
        if (r == 0) return (int)c;
        if (r < (uint32_t)MINSEG) return (int)(c - r);
        if ((uint32_t)KB - r < (uint32_t)MINSEG) return (int)(c + ((uint32_t)KB - r));
        return (int)c;
    };
    const int u0 = cut(w), u1 = cut(w + 1);
    if (u0 >= u1) return;
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

    // current segment: tile t, k blocks [kb_b, kb_e); the next one (always a tile start) is set up a segment ahead
    int t = u0 / KB;
    int kb_b = u0 - t * KB;
    int kb_e = min(KB, u1 - t * KB);
    int m0, n0, m0n, n0n;
    tile_origin(t, m0, n0);
    bool have_next = t * KB + kb_e < u1;
    if (have_next) tile_origin(t + 1, m0n, n0n);
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
            const uint8_t *st = smem + (gb & 1) * Cfg::STAGE_BYTES;
            const uint8_t *sn = smem + ((gb & 1) ^ 1) * Cfg::STAGE_BYTES;
            // whose blocks the refill slots of this k block fetch: blocks kb+1 (head part) and kb+2 (tail part) of this segment, or
            // blocks 0 / 1 of the next one (which starts its tile)
            const bool hn = kb + 1 >= kb_e, tn = kb + 2 >= kb_e;
            const v4i ha = hn ? a_rsrc_n : a_rsrc, hb = hn ? b_rsrc_n : b_rsrc;
            const v4i ta = tn ? a_rsrc_n : a_rsrc, tb = tn ? b_rsrc_n : b_rsrc;
            const float *hs = hn ? sc_src_n : sc_src, *ts = tn ? sc_src_n : sc_src;
            const int hk = hn ? kb + 1 - kb_e : kb + 1, tk = tn ? kb + 2 - kb_e : kb + 2;
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
        if (kb_b > 0) {
            // ---- a segment that does not start its tile (the first of this workgroup's run): the fp32 partial goes to this
            //      workgroup's slot, lane-linear (16 bytes per lane per accumulator tile), then the flag.  Every access to a slot or
            //      a flag carries sc1: it is served at the device-coherent level (a partial is read by a workgroup of another XCD),
            //      so neither side needs an L2 write-back or invalidate -- an acquire loop that invalidates the XCD's L2 on every
            //      poll took the operand panels of every workgroup of that XCD with it (first version: 2.3 x the run time)
            float *slot = sk.partials + (int64_t)w * SLOT + tid * 4;     // (one running pointer: an asm operand takes no immediate offset)
            if (!(sk.debug & 1))
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) {
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(slot), "v"(acc[mt][nt]) : "memory");
                    slot += NT * 4;
                    asm volatile("" : "+v"(slot));
                }
            wait_vmcnt<0>();                                        // this wave's rows have reached the coherent level ...
            barrier();                                              // ... every wave's have
            if (tid == 0) __hip_atomic_store(sk.flags + w, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (kb_e < KB) {
            break;      // a segment that starts its tile but does not end it is the LAST of the run: finished below, outside the loop
        } else {
            store_tile(acc, m0, n0);
        }
        if (!have_next) break;
        // ---- on to the next segment (a tile start): its descriptors become the current ones, the one after it is set up
        t += 1;
        kb_b = 0;
        kb_e = min(KB, u1 - t * KB);
        m0 = m0n; n0 = n0n;
        a_rsrc = a_rsrc_n; b_rsrc = b_rsrc_n; sc_src = sc_src_n;
        have_next = t * KB + kb_e < u1;
        if (have_next) {
            tile_origin(t + 1, m0n, n0n);
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
    if (kb_b == 0 && kb_e < KB) {
        // ---- the run ended inside a tile it started: add the partials of the workgroups that hold the rest of that tile, in k order
        //      (they were written at the START of those workgroups' runs), then store.  Out here the fragment and scale registers
        //      of the main loop are dead, so a whole row of accumulator tiles travels per round trip.
        int done = t * KB + kb_e;
        const int tile_end = (t + 1) * KB;
        int wq = w + 1;
        while (done < tile_end && wq < P) {
            const int c0 = cut(wq), c1 = cut(wq + 1);
            if (c1 > c0) {     // (a workgroup with an empty run holds nothing)
                if (!(sk.debug & 4))
                while (__hip_atomic_load(sk.flags + wq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(8);
                const float *slot = sk.partials + (int64_t)wq * SLOT + tid * 4;
                if (!(sk.debug & 2))
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    v4f q[TN];
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt) {
                        asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(q[nt]) : "v"(slot) : "memory");
                        slot += NT * 4;
                        asm volatile("" : "+v"(slot));
                    }
                    wait_vmcnt<0>();
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt) {
                        asm volatile("" : "+v"(q[nt]));     // (the load's result is valid from here on)
                        acc[mt][nt].x += q[nt].x; acc[mt][nt].y += q[nt].y; acc[mt][nt].z += q[nt].z; acc[mt][nt].w += q[nt].w;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the flag goes back to "not raised": a graph replay launches with the same epoch
                barrier();
                if (tid == 0) __hip_atomic_store(sk.flags + wq, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                done = min(tile_end, c1);
            }
            ++wq;
        }
        store_tile(acc, m0, n0);
    }
}

}  // namespace dga
