// Stream-K form of the bf16-exact policy's persistent 128 x 256 kernel (gemm_fp8_bf16x_persistent_kernel.hpp; dispatchPolicyTag 7,
// kernelSerial DGA_KERNEL_STREAMK_ONE_LAUNCH): ONE launch, one workgroup per CU.  The whole rounds of a dense raster run as in the
// persistent kernel; the LAST, partial round -- R tiles for P workgroups -- is cut along K so that it costs a fraction of a round:
//   * R <= P / 2: every remainder tile is cut into s = min(P / R, 8) equal k ranges (part-major over the workgroups, so that the
//     workgroups of an XCD walk the same k slice of the panels they share, as in a whole round);
//   * P / 2 < R < P: R workgroups ("mains") multiply k blocks [0, k*) of one tile each, the P - R spare workgroups take the tails
//     [k*, KB) of t = ceil(R / (P - R)) tiles each, one after the other, with k* = KB t / (t + 1): both kinds finish together after
//     ~R / P of a round, and every workgroup of a kind is at the same k offset at the same time.  (Cutting the raster's k blocks
//     into P contiguous runs instead puts every workgroup at its own k offset: the L2 stops serving the panels,
//     profiles/r05_streamk_skewed_ab.txt.)
// A piece that is not the ADDING piece of its tile leaves its fp32 accumulators in the workspace (128 KB, one slot per piece,
// write-through stores, then a flag); the adding piece -- the last k range in the first form, the main in the second -- waits for
// the flags of its tile, adds the partials in k order (a fixed order: the result does not depend on timing) and stores the bf16
// rows.  Only the adding piece waits, at the very end of its run, for pieces that finish at the same time or earlier.
// Arithmetic of a piece = the policy's (exact conversions, four chained bf16 MFMAs per scale block, fp32 promotion); a tile that is
// cut has its k blocks summed in two or more fp32 chains that are then added -- the split-K builds' arithmetic, not the one-launch
// bits (tests/test_bf16x_streamk_gpu.py: the policy's bar against the oracle, determinism, graph replay, every cut).
// Counterpart in the reference: kernel type 4, PaddingStreamkMatmulKernel -- Stream-K split of the k loop over all cores +
// StreamkReduceAdd over fp32 partials (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_streamk_matmul_kernel.h:94-98;
// selection rule op_host/op_tiling/select_kernel.cpp:303-331).
// A flag is "raised" when it holds this launch's 64-bit epoch; its one reader puts it back to 0 (nothing to zero in front of a launch,
// captured or not: a replay repeats the epoch and finds zeros).  Every workgroup must be resident at once (the launcher sizes the grid for that).
#pragma once
#include "gemm_fp8_kernel.hpp"
#include "gemm_fp8_streamk_kernel.hpp"   // StreamKArgs

namespace dga {

// The cut of the partial round, computed the same way on the host (selector, launcher, tests) and in the kernel.
struct BxStreamKPlan {
    int n_dp;        // whole tiles per workgroup
    int R;           // tiles of the partial round (0: none)
    int form;        // 0: no cut (R == 0, or a cut would leave a piece shorter than two k blocks); 1: s equal ranges; 2: mains + tails
    int s;           // form 1: ranges per tile
    int t;           // form 2: tails per spare workgroup
    int kstar;       // form 2: first k block of a tail
};
__host__ __device__ inline BxStreamKPlan bx_streamk_plan(int tiles, int P, int KB)
{
    BxStreamKPlan pl{};
    pl.n_dp = tiles / P;
    pl.R = tiles - pl.n_dp * P;
    pl.form = 0; pl.s = 1; pl.t = 0; pl.kstar = KB;
    if (pl.R == 0) return pl;
    if (2 * pl.R <= P) {
        int s = P / pl.R;
        if (s > 8) s = 8;
        while (s > 1 && KB / s < 2) --s;
        if (s > 1) { pl.form = 1; pl.s = s; }
    } else {
        const int spare = P - pl.R;
        const int t = (pl.R + spare - 1) / spare;
        int ks = (KB * t + (t + 1) / 2) / (t + 1);
        if (ks > KB - 2) ks = KB - 2;
        if (ks >= 2) { pl.form = 2; pl.t = t; pl.kstar = ks; }
    }
    return pl;
}

template <bool KTAIL>
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
gemm_fp8_bf16x_streamk_kernel(const GemmParams p, const StreamKArgs sk)
{
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN, TM = Cfg::TM, TN = Cfg::TN, DNT = Cfg::DNT;
    constexpr int NL = Cfg::LOADS_PER_STAGE, TILES = TM * TN, G = 4 * TM, LAGT = 2, RING = 4, NT = Cfg::NT;
    constexpr int SLOT = BM * BN;      // floats of a partial tile
    static_assert(Cfg::NT == 512 && DNT == 512 && TILES % RING == 0 && 4 * TILES >= 4 + NL && 16 % G == 0, "the MATH = 1 schedule");
    typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 15, kg = lane >> 4;
    const int KB = p.kb_n;   // >= 2 (host)

    // ---- this workgroup's segments: its whole tiles (the XCD's contiguous chunk of the whole rounds, strided by the XCD's workgroups,
    //      as in the persistent kernel), then its piece(s) of the partial round
    const int P = gridDim.x;
    const int tiles = p.tiles_m * p.tiles_n;
    const BxStreamKPlan pl = bx_streamk_plan(tiles, P, KB);
    const int full = pl.n_dp * P;
    int first = 0, step = P, local = blockIdx.x, q = blockIdx.x;
    if (p.xcd_remap && (P & 7) == 0) {
        const int xcd = blockIdx.x & 7;
        first = xcd * (full >> 3);
        step = P >> 3;
        local = blockIdx.x >> 3;
        q = xcd * (P >> 3) + local;                // logical index: the workgroups of an XCD are neighbours
    }
    // pieces of the partial round, by logical index q
    //   form 0: q < R takes tile q whole.
    //   form 1: piece q = j * R + r (part-major) for q < R * s: k range j of tile r; the last range adds.
    //   form 2: P - R spares (spare i: the tails of tiles i, i + used, ... in turn) and R mains (k blocks [0, k*) of one tile each; a main adds), both kinds spread over the XCDs.
    int n_pieces = 0, piece_r0 = 0, piece_stride = 1, piece_kb0 = 0, piece_kb1 = KB, piece_role = 0;   // role 0: whole tile; 1: leaves a partial; 2: adds
    if (pl.R > 0) {
        if (pl.form == 0) {
            if (q < pl.R) { n_pieces = 1; piece_r0 = q; }
        } else if (pl.form == 1) {
            if (q < pl.R * pl.s) {
                const int j = __builtin_amdgcn_readfirstlane(q / pl.R);
                n_pieces = 1; piece_r0 = q - j * pl.R;
                piece_kb0 = (j * KB) / pl.s; piece_kb1 = ((j + 1) * KB) / pl.s;
                piece_role = j == pl.s - 1 ? 2 : 1;
            }
        } else {
            // mains and spares are dealt EVENLY over the XCDs (an XCD whose 32 CUs are all mains ran its k blocks 17 % slower than one
            // that holds 25: profiles/r06_bx_streamk.txt): XCD x holds the mains of its contiguous share of the R tiles on its first
            // workgroups and spares on the rest
            const int spare = P - pl.R;
            int main_r = -1, spare_i = -1;           // this workgroup: main of tile main_r, or spare number spare_i
            if (p.xcd_remap && (P & 7) == 0) {
                const int xcd = blockIdx.x & 7, per = P >> 3, qr = pl.R >> 3, rr = pl.R & 7;
                const int cnt = qr + (xcd < rr ? 1 : 0);                       // mains of this XCD
                const int first_r = xcd < rr ? xcd * (qr + 1) : rr * (qr + 1) + (xcd - rr) * qr;
                if (local < cnt) main_r = first_r + local;
                else spare_i = xcd * per - first_r + (local - cnt);           // spares of the XCDs before this one + its own
            } else {
                if (q < spare) spare_i = q; else main_r = q - spare;
            }
            if (spare_i >= 0) {
                // spare i walks the tails of tiles i, i + used, i + 2 used, ...: at any moment the spares are on NEIGHBOURING tiles
                // (shared operand panels), as the mains are
                piece_stride = (pl.R + pl.t - 1) / pl.t;      // spares in use
                piece_r0 = spare_i;
                n_pieces = spare_i < piece_stride ? (pl.R - spare_i + piece_stride - 1) / piece_stride : 0;
                piece_kb0 = pl.kstar; piece_kb1 = KB; piece_role = 1;
            } else {
                n_pieces = 1; piece_r0 = main_r; piece_kb0 = 0; piece_kb1 = pl.kstar; piece_role = 2;
            }
        }
    }
#ifdef DGA_BXSK_KNOBS      // (diagnostic builds: time one kind of piece alone -- results are then wrong; p.tail_begin carries the knob)
    if ((p.tail_begin & 1) && piece_role == 1) n_pieces = 0;                 // no partial is written ...
    if ((p.tail_begin & 2) && piece_role == 2) n_pieces = 0;
    if ((p.tail_begin & 5) && piece_role == 2) piece_role = 0;               // ... so nobody may wait for one
#endif
    const int n_seg = pl.n_dp + n_pieces;
    if (n_seg == 0) return;
    struct Seg { int m0, n0, kb0, kb1, role, r; };
    auto seg_at = [&](int i, Seg &s) {
        const int t_in = i < pl.n_dp ? first + local + i * step : full + piece_r0 + (i - pl.n_dp) * piece_stride;
        const int gm = p.raster_group;
        const int per = gm * p.tiles_n;
        const int band = t_in / per;
        const int row0 = band * gm;
        const int rows = min(p.tiles_m - row0, gm);
        const int loc = t_in - band * per;
        s.m0 = __builtin_amdgcn_readfirstlane((row0 + loc % rows) * BM);
        s.n0 = __builtin_amdgcn_readfirstlane((loc / rows) * BN);
        s.kb0 = i < pl.n_dp ? 0 : piece_kb0;
        s.kb1 = i < pl.n_dp ? KB : piece_kb1;
        s.role = i < pl.n_dp ? 0 : piece_role;
        s.r = i < pl.n_dp ? 0 : piece_r0 + (i - pl.n_dp) * piece_stride;
    };

    // ---- LDS-DMA sources (gemm_fp8_bf16x_persistent_kernel.hpp)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    constexpr uint32_t kOutOfRange = 0x80000000u;
    const int a_col = ((tid & 7) ^ swz_a(tid >> 3)) * 16;
    const int b_col = ((tid & 7) ^ swz_b(tid >> 3)) * 16;
    auto a_off = [&](const Seg &t, int it) -> uint32_t {
        const int row = (it * DNT + tid) >> 3;
        return row < p.m - t.m0 ? (uint32_t)row * (uint32_t)p.lda + a_col : kOutOfRange;   // rows at or beyond M: zero-filled, not fetched
    };
    auto b_off = [&](const Seg &t, int it) -> uint32_t {
        const int row = (it * DNT + tid) >> 3;
        return (uint32_t)min(row, p.n - 1 - t.n0) * (uint32_t)p.ldb + b_col;
    };
    auto sc_ptr = [&](const Seg &t) -> const float * {   // slot tid: [0, BM) sfa rows of the tile, then its sfb blocks
        return tid < BM ? p.sfa + (int64_t)min(t.m0 + tid, p.m - 1) * p.sfa_ld
                        : p.sfb + (int64_t)min(t.n0 / 128 + min(tid - BM, 7), p.nb_n - 1) * p.kb_n;
    };
    auto a_desc = [&](const Seg &t) { return make_rsrc(p.a + (int64_t)t.m0 * p.lda, (int64_t)(p.m - t.m0) * p.lda); };
    auto b_desc = [&](const Seg &t) { return make_rsrc(p.b + (int64_t)t.n0 * p.ldb, (int64_t)(p.n - t.n0) * p.ldb); };
    static_assert(Cfg::SC_ITERS == 1, "one scale piece per stage");

    Seg T{}, Tn{};
    int seg = 0;
    seg_at(0, T);
    v4i a_rsrc = a_desc(T), b_rsrc = b_desc(T), a_rsrc_n = a_rsrc, b_rsrc_n = b_rsrc;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
    const float *sc_src;
    auto load_offsets = [&](const Seg &t) {
#pragma unroll
        for (int it = 0; it < Cfg::A_ITERS; ++it) a_voff[it] = a_off(t, it);
#pragma unroll
        for (int it = 0; it < Cfg::B_ITERS; ++it) b_voff[it] = b_off(t, it);
        sc_src = sc_ptr(t);
    };
    load_offsets(T);
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
    // the same from the NEXT segment's k block kb (no next segment: every lane out of range -- zeros land, nothing is fetched; the
    // scale piece re-reads the current segment's last block)
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
            dma4(has_next ? sc_ptr(Tn) + kb : sc_src + (T.kb1 - 1), lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES + wave * 256);
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
    auto rows_present = [&](const Seg &t) { return t.m0 + wm * (BM / Cfg::kWM) < p.m; };   // (wave-uniform)

    // ---- prologue: the first two blocks of the first segment on their way, the first landed
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_cur(idx, d, T.kb0 + d);
    wait_vmcnt<NL>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    clear_tile();
    bool active = rows_present(T);
    if (active) first_fragments(smem);
    int cur = 0, nxt = 1, fill = 2;
    int flag_pending = -1;     // slot of a partial whose flag is still to be raised

    // one k block; LAST2: the refill is the next segment's block (kb + 2 - kb1) blocks behind its first
    auto k_block = [&](int kb, auto last2c) __attribute__((always_inline)) {
        constexpr bool LAST2 = decltype(last2c)::value;
        auto refill = [&](int idx) {
            if constexpr (LAST2) issue_next(idx, fill, kb + 2 - T.kb1 + Tn.kb0);
            else issue_cur(idx, fill, kb + 2);
        };
        wait_vmcnt<0>();                         // this wave's pieces of the next block (issued a block ago) have landed
        __builtin_amdgcn_s_barrier();            // ... everyone's have; and everyone has left the block whose stage is refilled now
        asm volatile("" ::: "memory");
        if (flag_pending >= 0) {                 // (the previous segment's partial: every wave's stores are drained, see the boundary)
            if (tid == 0) __hip_atomic_store(sk.flags + flag_pending, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag_pending = -1;
        }
        if (!active) {
#pragma unroll
            for (int idx = 0; idx < NL; ++idx) refill(idx);
        } else {
            const uint8_t *sc = smem + cur * Cfg::STAGE_BYTES;   // being consumed (B raw reloads of this block)
            const uint8_t *sn = smem + nxt * Cfg::STAGE_BYTES;   // landed: the next block's fragments are read ahead from it
#pragma unroll
            for (int u = 0; u < 4 * TILES; ++u) {
                const int t = u >> 2, q4 = u & 3, nt = t / TM, mt = t % TM, g = u % G;
                part[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(v8bf, bfx[nt & 1][q4]), __builtin_bit_cast(v8bf, afx[mt][q4]),
                    q4 == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t % RING], 0, 0, 0);
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
                if (nt == TN - 1 && q4 == 0) {
                    araw[mt & 1][0] = *(const v4i *)(sn + a_off0 + mt * 2048);
                    araw[mt & 1][1] = *(const v4i *)(sn + a_off1 + mt * 2048);
                }
                if (nt == TN - 1 && mt >= 1) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(mt - 1) & 1], afx[mt - 1], 4 * q4 + c);
                }
                if (t == 0) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(TM - 1) & 1], afx[TM - 1], 4 * q4 + c);
                }
                if (u == 4 * TILES - 8) {
                    const float sfbn = *(const float *)(sn + sb_off);
#pragma unroll
                    for (int i = 0; i < TM; ++i) s_nxt[i] = *(const float *)(sn + sa_off + i * 64) * sfbn;
                }
                {
                    const int j = t >= LAGT ? t - LAGT : TILES + t - LAGT, jn = j / TM, jm = j % TM;
                    const float sv = t >= LAGT ? s_cur[jm] : s_old[jm];
                    acc[jm][jn][q4] = __builtin_fmaf(part[j % RING][q4], sv, acc[jm][jn][q4]);
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

    auto store_tile = [&]() {
        uint16_t *C = p.out;
        const int m_row = T.m0 + wm * (BM / Cfg::kWM) + li;
        const int n_base = T.n0 + wn * (BN / WN) + 8 * kg;
        const bool vec_ok = ((p.ldc & 7) == 0) && ((((uintptr_t)C) & 15) == 0);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m_row + mt * 16;
            if (m >= p.m) continue;
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
                    if (p.out_nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(crow + n), "v"(pk) : "memory");
                    else *(v4i *)(crow + n) = pk;
                } else {
                    const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8)
                        if (n + q8 < p.n) crow[n + q8] = e[q8];
                }
            }
        }
    };

    for (;;) {
        // the segment after this one (its descriptors are needed from this segment's second-to-last k block on)
        has_next = seg + 1 < n_seg;
        if (has_next) { seg_at(seg + 1, Tn); a_rsrc_n = a_desc(Tn); b_rsrc_n = b_desc(Tn); }
        for (int kb = T.kb0; kb < T.kb1 - 2; ++kb) k_block(kb, std::false_type{});
        k_block(T.kb1 - 2, std::true_type{});
        k_block(T.kb1 - 1, std::true_type{});

        // ---- boundary: the last LAGT tiles of the last block; then the tile's rows (whole tile / adding piece) or its partial
        if (active) {
#pragma unroll
            for (int t = 0; t < LAGT; ++t) {
                const int j = TILES + t - LAGT, jn = j / TM, jm = j % TM;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) acc[jm][jn][q4] = __builtin_fmaf(part[j % RING][q4], s_old[jm], acc[jm][jn][q4]);
            }
        }
        if (T.role == 2) break;     // the adding piece is a workgroup's LAST segment: finished below, outside the loop (its fragments are dead there)
        if (T.role == 1) {
            // this piece's fp32 accumulators go to its slot, lane-linear (16 bytes per lane per accumulator tile), write-through; then
            // the flag, behind every wave's drained stores (cdna_hip_programming.md Guideline 16, the sc1 form)
            const int slot = pl.form == 1 ? q : T.r;
            if (active) {
                float *dst = sk.partials + (int64_t)slot * SLOT + tid * 4;
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt) {
                        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(dst), "v"(acc[mt][nt]) : "memory");
                        dst += NT * 4;
                        asm volatile("" : "+v"(dst));
                    }
            }
            if (has_next) {
                // the flag goes up behind the next segment's first barrier: every wave waits vmcnt(0) in front of it for its DMA anyway,
                // which drains these stores too -- no wait of its own
                flag_pending = slot;
            } else {
                wait_vmcnt<0>();                                    // this wave's rows have reached the coherent level ...
                __builtin_amdgcn_s_barrier();                       // ... every wave's have
                asm volatile("" ::: "memory");
                if (tid == 0) __hip_atomic_store(sk.flags + slot, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        } else if (active) {
            store_tile();
        }
        if (!has_next) break;
        // ---- the next segment becomes the current one: its first block sits in stage `cur`, its second is on its way into `nxt`
        const bool was_active = active;
        T = Tn; seg += 1;
        a_rsrc = a_rsrc_n; b_rsrc = b_rsrc_n;
        load_offsets(T);
        active = rows_present(T);
        clear_tile();
        if (active && !was_active) first_fragments(smem + cur * Cfg::STAGE_BYTES);
    }
    wait_vmcnt<0>();   // the refills past the last segment (zeros) land in LDS nobody reads: drain them before exit
    if (T.role == 2) {
        // ---- the adding piece: the other pieces of its tile (form 1: the s - 1 lower k ranges; form 2: the tail), summed in k order,
        //      then this piece's own accumulators (two pieces: the order of one addition does not matter; more: the others' sum is a
        //      k-ordered chain and this piece is the last range of form 1).  Read with sc1 loads (cdna_hip_programming.md Guideline 16).
        const int n_other = pl.form == 1 ? pl.s - 1 : 1;
        // (one thread watches the flags -- each has this ONE reader -- and puts them back to 0 behind the barrier: a launch replayed from
        //  a graph repeats its epoch and finds the flags as an ordinary launch does, no memset node in front of the kernel)
        if (tid == 0)
            for (int j = 0; j < n_other; ++j) {
                const int slot = pl.form == 1 ? j * pl.R + T.r : T.r;
                while (__hip_atomic_load(sk.flags + slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) __builtin_amdgcn_s_sleep(8);
            }
        __syncthreads();
        if (tid == 0)
            for (int j = 0; j < n_other; ++j)
                __hip_atomic_store(sk.flags + (pl.form == 1 ? j * pl.R + T.r : T.r), 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (active) {
            constexpr int CH = TILES / 2;      // accumulator tiles per round trip: eight 16-byte loads in flight per lane
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                v4f v[CH];
                for (int j = 0; j < n_other; ++j) {
                    const int slot = pl.form == 1 ? j * pl.R + T.r : T.r;
                    const float *src = sk.partials + (int64_t)slot * SLOT + (h * CH * NT + tid) * 4;
                    v4f w[CH];
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(w[i]) : "v"(src) : "memory");
                        src += NT * 4;
                        asm volatile("" : "+v"(src));
                    }
                    wait_vmcnt<0>();
#pragma unroll
                    for (int i = 0; i < CH; ++i) {
                        asm volatile("" : "+v"(w[i]));     // (the loads' results are valid from here on)
                        if (j == 0) v[i] = w[i];
                        else { v[i].x += w[i].x; v[i].y += w[i].y; v[i].z += w[i].z; v[i].w += w[i].w; }
                    }
                }
#pragma unroll
                for (int i = 0; i < CH; ++i) {
                    const int idx = h * CH + i, mt = idx / TN, nt = idx % TN;
                    acc[mt][nt].x = v[i].x + acc[mt][nt].x; acc[mt][nt].y = v[i].y + acc[mt][nt].y;
                    acc[mt][nt].z = v[i].z + acc[mt][nt].z; acc[mt][nt].w = v[i].w + acc[mt][nt].w;
                }
            }
            store_tile();
        }
    }
}

}  // namespace dga
