// FP8 (OCP e4m3fn) block-scaled NT GEMM for gfx950 (MI355X / CDNA4).
//
//   out[m,n] = bf16( sum_kb  sfa[m,kb] * sfb[n/128,kb] * sum_{k in kb} A[m,k] * B[n,k] )
//
// This is the CDNA4 counterpart of the reference's device K-loop
// (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:123-369,
//  same logic in .../deep_gemm_ascend/include/impls/mmad_jit.cpp:114-415):
//   GM -> L1 (double buffered)   ==>  HBM -> LDS by LDS-DMA (global_load_lds, 2 stages)
//   L1 -> L0A/L0B                ==>  LDS -> VGPR fragments (ds_read_b128, XOR-swizzled image)
//   Mmad into fp32 L0C           ==>  v_mfma_f32_16x16x128_f8f6f4 (one MFMA = one 128-wide scale block)
//   (new) two-level dequant      ==>  acc += partial * (sfa*sfb), fp32 FMA in registers
//   Fixpipe L0C -> GM            ==>  fp32 -> bf16 (v_cvt_pk_bf16_f32) + 16-byte row stores
//
// Orientation.  The MFMA computes D = Aop . Bop with D[i][j] held as
// (row i = 4*(lane>>4)+r, col j = lane&15).  We feed the *B matrix tile* as Aop
// (rows = n) and the *A matrix tile* as Bop (cols = m), so that every lane owns
// 4 consecutive n of ONE output row m: the 1x128 A-scale is then one value per
// lane per m-tile, and two n-tiles give 8 consecutive bf16 = one 16-byte store.
// The n index that MFMA row i of n-tile nt stands for is
//     n_local(nt, i) = 32*(nt>>1) + 8*(i>>2) + 4*(nt&1) + (i&3).
// K placement inside an MFMA is symmetric in Aop/Bop, so both fragments take
// bytes [16*kg,+16) and [64+16*kg,+16) of the 128-byte k block (kg = lane>>4).
#pragma once
#include "dga_device_common.hpp"
#include "dga_hip.h"

namespace dga {

struct GemmParams {
    const uint8_t *a;      // [G][m_rows][lda] e4m3fn bytes
    const float *sfa;      // [G][m_rows][kb_n]
    const uint8_t *b;      // [G][n][ldb]
    const float *sfb;      // [G][nb_n][kb_n]
    uint16_t *out;         // [G][m_rows][ldc] bf16 bits
    const int32_t *masked_m;  // grouped: device int32[G]; dense: nullptr
    const int32_t *m_indices;  // contiguous-grouped: device int32[m], the B group of every row (-1 = skip); else nullptr
    const int64_t *row_index;  // indexed masked-grouped: device int64[groups * m]; row r of group g is row row_index[g*m + r]
                               // of ONE flat source / destination (a, sfa and out then have group stride 0): the kernel
                               // gathers token rows where they lie and scatters the result rows in place of a pack / unpack
    int64_t sfa_ld;            // floats between consecutive sfa rows (kb_n unless the scales ride inside payload rows)
    int64_t a_bytes;           // indexed: byte extent of the flat A source (bounds the buffer descriptor)
    int b_groups;              // contiguous-grouped: number of B groups (bounds the device-side index)
    int m;                 // dense: M; grouped: m_max (rows allocated per group)
    int n, k, kb_n, nb_n;
    int64_t lda, ldb, ldc;       // row strides in elements
    int64_t a_gs, b_gs, c_gs;    // group strides in elements
    int64_t sfa_gs, sfb_gs;
    int tiles_m, tiles_n, groups;
    int raster_group;            // tile-rows walked together (swizzleOffset analogue, tiling_params.h:63)
    int xcd_remap;               // 1: contiguous tile chunk per XCD (blocks b, b+8, ... share an XCD)
    float *partial;              // split-K: fp32 slabs [splitk][m][n] in the caller's workspace (dense only)
    int splitk, kb_per_split;    // splitk > 1: block -> (split, tile); split s covers k blocks [s*kb_per_split, +kb_per_split)
    int b_nt;                    // persistent builds: B fetched with the non-temporal policy: 0 never, 1 always, 2 per tile
                                 // by its row count (the masked grouped stream: weights read once by one CU)
    int out_nt;                  // persistent builds: bf16 rows stored with the non-temporal policy (the masked grouped stream)
    int launch_tiles;            // > 0: grid size of this launch (the first launch_tiles tiles of the raster); 0: all tiles
    int tail_begin, tail_sub;    // tail_sub = 2: this launch's tiles are QUARTER tiles (2 x 2 per parent) of the parent
                                 // raster's tiles [tail_begin, ...); tiles_m / tiles_n then hold the PARENT raster
    unsigned long long *stamps;  // diagnostics only: per-wave segment cycle sums (-DDGA_STAMPS builds, 8 words per wave) or the
                                 // loop clock of the CLK = true instantiations (2 words per wave); nullptr in product calls
};

// In-kernel stamps (diagnostic build only; cdna_hip_programming.md section 7 "In-kernel stamps").
#ifdef DGA_STAMPS
#define DGA_STAMP_DECL unsigned long long st_prev = 0, st_seg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DGA_STAMP_START()                                                                       \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_prev)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#define DGA_STAMP(i)                                                                            \
    do {                                                                                        \
        unsigned long long st_now;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        st_seg[i] += st_now - st_prev;                                                          \
        st_prev = st_now;                                                                       \
    } while (0)
#define DGA_STAMP_CLOCK(slot_t, slot_rt)                                                        \
    do {                                                                                        \
        unsigned long long c0, c1;                                                              \
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(c1)::"memory"); \
        st_seg[slot_t] = c0 - st_seg[slot_t];                                                   \
        st_seg[slot_rt] = c1 - st_seg[slot_rt];                                                 \
    } while (0)
#define DGA_STAMP_ABS(i)   /* absolute 100 MHz real-time stamp into slot i (wave entry / loop start / loop end / wave exit) */ \
    do {                                                                                        \
        unsigned long long st_abs;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_abs)::"memory");      \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        st_seg[i] = st_abs;                                                                     \
    } while (0)
#define DGA_STAMP_FLUSH()                                                                       \
    do {                                                                                        \
        if (p.stamps && lane == 0)                                                              \
            for (int q = 0; q < 8; ++q) p.stamps[((size_t)blockIdx.x * (NT / 64) + wave) * 8 + q] = st_seg[q]; \
    } while (0)
#else
#define DGA_STAMP_DECL
#define DGA_STAMP_CLOCK(a, b) do { } while (0)
#define DGA_STAMP_START() do { } while (0)
#define DGA_STAMP_ABS(i) do { } while (0)
#define DGA_STAMP(i) do { } while (0)
#define DGA_STAMP_FLUSH() do { } while (0)
#endif


// Clock of the main loop (MI355X_MICROARCH.md "DVFS give-back" item 6): shader ticks / 100 MHz real-time ticks, stamped
// once in front of and once behind the k loop.  Compiled in only in the CLK = true instantiations, which the
// dga_gemm_fp8_loop_clock() diagnostic launches; the product kernels (CLK = false) carry no stamp.  The two words per
// wave go to a buffer nothing else reads.
template <bool ON>
struct LoopClock {
    unsigned long long t = 0, rt = 0;
    __device__ __forceinline__ void tick()
    {
        if constexpr (ON) {
            unsigned long long c0, c1;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c0), "=s"(c1)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            t = c0 - t;
            rt = c1 - rt;
        }
    }
    __device__ __forceinline__ void flush(unsigned long long *dst, int slot, int lane) const
    {
        if constexpr (ON) {
            if (dst && lane == 0) { dst[2 * slot] = t; dst[2 * slot + 1] = rt; }
        }
    }
};

// PP = 0: every wave runs the same k-block loop (one barrier per k block).
// PP = 1 ("ping-pong", 256x256 tile with 8 waves only): the second-dispatched half of the workgroup runs half a k
//         block behind the first, so one half's LDS fragment burst always overlaps the other half's MFMAs.
// MATH = 0: the fp8 matrix instruction (one v_mfma_scale_f32_16x16x128_f8f6f4 per 128-wide scale block).
// MATH = 1 (PP = 0, three LDS stages): the bf16-exact policy -- the e4m3 bytes are up-converted to bf16 in registers (exact)
//         and a scale block is four chained v_mfma_f32_16x16x32_bf16, whose sums are fp32-class (see the loop below).
// MATH = 2 (PP = 0 or 2): the fp8 matrix instruction with the block scales in its E8M0 operands -- for scale tensors whose values
//         are exact powers of two ("UE8M0" scales, what upstream DeepGEMM's per_token_cast_to_fp8(..., use_ue8m0=True) writes):
//         acc = v_mfma_scale_f32_16x16x128_f8f6f4(b, a, acc, e8m0(sfb), e8m0(sfa)) -- the MFMA accumulates in place exactly as
//         the reference's Mmad(c1Local, ..., init on first) does (generate_code.hpp:320-335); no partial-sum ring, no promotion
//         FMAs: the vector pipe carries nothing but the matrix instruction.  C-in keeps fp32 (scripts/ubench/probe_mfma_scale_acc.hip
//         -> profiles/r05_probe_scale_acc.txt: 13 directed cases = fp32(C + p); K = 4096 / 7168 chains give the SAME bf16 output as the
//         promotion form on all 76 800 outputs, max fp32 difference 2^-27 S).
// MATH = 3 (PP = 0, three LDS stages): the bf16-exact arithmetic for power-of-two scales.  v_cvt_scalef32_pk_bf16_fp8 multiplies by a
//         power of two for free, so the A fragments are converted with sfa[m, kb] * sfb[n / 128, kb] folded in (exact: an e4m3 value
//         times a power of two is a bf16 value) and the bf16 MFMA chain simply runs on through every k block with C-in -- exact
//         products, fp32-class sums, NO promotion: per MFMA the vector pipe carries conversions only, and since no vector instruction
//         touches an accumulator they may live in AGPRs: four waves with 64 x 128 wave tiles (1.5 conversions per MFMA instead of
//         2 + a promotion FMA).  Same LDS image, DMA and fragment reads as MATH = 1.
// UNAL (loader-wave builds, plain loop, dense): the operands' rows start at ANY byte (K % 16 != 0, no padded copy): the loader waves
//         fetch them to registers with dword-aligned loads, realign (v_alignbyte), zero the bytes beyond K and ds_write the same
//         swizzled image the LDS-DMA would have written -- the computing waves are unchanged.  The counterpart of the reference's
//         PaddingCommon kernel, which fuses the re-layout with the matmul
//         (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_common_matmul_kernel.h:33-107).
template <class Cfg, int PP, bool KTAIL, bool CLK = false, int MATH = 0, bool UNAL = false>
__global__ void __launch_bounds__(Cfg::NT) gemm_fp8_blockscaled_nt_kernel(const GemmParams p)
{
    static_assert(!UNAL || (Cfg::kLC && PP == 0 && MATH == 0 && KTAIL), "unaligned rows: loader waves, plain loop, fp8 matrix instruction");
    static_assert(MATH != 2 || PP != 1, "hardware-scale builds: plain or continuous loop");
    static_assert(MATH != 3 || PP == 0, "bf16-exact builds: plain loop");
    LoopClock<CLK> loop_clock;
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN;
    constexpr int NT = Cfg::NT, TM = Cfg::TM, TN = Cfg::TN;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    // loader / consumer builds: waves [0, WM*WN) compute, waves [WM*WN, 2*WM*WN) only issue the LDS-DMA
    constexpr bool LC = Cfg::kLC;
    const bool loader = LC && wave >= Cfg::kWM * WN;
    const int dwave = loader ? wave - Cfg::kWM * WN : wave;   // this wave's 1 KiB slot in every DMA piece
    static_assert(!LC || PP == 0, "the loader waves ride on the plain loop");
    static_assert(!LC || Cfg::DNT * 1 == Cfg::DMA_WAVES * 64, "loader waves carry the whole DMA");

    // ---- tile id: XCD-aware remap (blocks b, b+8, ... share an XCD and its L2), then
    //      a grouped raster so that an XCD's consecutive tiles share A and B panels.
    int nwg = gridDim.x, bid = blockIdx.x;
    // contiguous-grouped layout with tiles taller than the segment alignment: the grid is doubled, the second copy
    // of a tile ("pass 1") only works when the tile straddles two groups (see below)
    int pass = 0;
    if (BM > DGA_CONTIGUOUS_M_ALIGNMENT && p.m_indices) {
        nwg >>= 1;
        if (bid >= nwg) { pass = 1; bid -= nwg; }
    }
    int tile;
    {
        const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = p.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
    }
    const int tiles_per_group = p.tiles_m * p.tiles_n;
    // split-K (dense only, groups == 1): the grid is splitk x tiles, split-major, so the splits of one tile run on
    // different XCDs/CUs at the same time; grouped: group-major
    const int split = p.splitk > 1 ? tile / tiles_per_group : 0;
    // (a tail launch is dense: its grid counts QUARTER tiles, which outnumber the parent raster's tiles once the tail is longer than a
    //  quarter of it -- read as a group index that sent the stores of such a launch past the output, a GPU memory fault)
    const int g = (p.splitk > 1 || p.tail_sub) ? 0 : tile / tiles_per_group;
    int t_in = tile - (p.splitk > 1 ? split : g) * tiles_per_group;
    int tm, tn;
    const int sub = p.tail_sub ? (t_in & 3) : 0;
    if (p.tail_sub) t_in = p.tail_begin + (t_in >> 2);
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
    if (p.tail_sub) {  // quarter tile (sub & 1, sub >> 1) of parent tile (tm, tn)
        tm = 2 * tm + (sub & 1);
        tn = 2 * tn + (sub >> 1);
    }
    const int M = p.masked_m ? min(p.masked_m[g], p.m) : p.m;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;  // empty expert / fully masked tile: nothing read, nothing written
    if (p.tail_sub && n0 >= p.n) return;  // a quarter tile beyond the matrix edge
    // contiguous-grouped layout: one A/out matrix, the B group comes from the index of the tile's first row (the
    // layout contract aligns group segments to the tile height); padding tiles (index -1) do nothing
    //   A tile taller than the alignment (256 rows = two 128-row blocks) may hold two groups: pass 0 computes the
    //   tile against the first block's group, pass 1 -- a second workgroup -- against the second block's group if
    //   that differs; each stores only the rows whose index matches (epilogue).  Uniform tiles cost pass 1 two loads.
    int bg = g;
    if (p.m_indices) {
        bg = p.m_indices[m0];
        if (BM > DGA_CONTIGUOUS_M_ALIGNMENT) {
            const int m1 = m0 + DGA_CONTIGUOUS_M_ALIGNMENT;
            const int g1 = m1 < M ? p.m_indices[m1] : bg;
            if (pass == 1) {
                if (g1 == bg) return;
                bg = g1;
            }
        }
        if (bg < 0 || bg >= p.b_groups) return;
    }
    const int kb_begin = p.splitk > 1 ? split * p.kb_per_split : 0;
    const int kb_end = p.splitk > 1 ? min(p.kb_n, kb_begin + p.kb_per_split) : p.kb_n;

    const int64_t *ridx = p.row_index ? p.row_index + (int64_t)g * p.m : nullptr;   // slot -> row of the flat buffers
    const uint8_t *A = p.a + (int64_t)g * p.a_gs;
    const uint8_t *B = p.b + (int64_t)bg * p.b_gs;
    const float *SFA = p.sfa + (int64_t)g * p.sfa_gs;
    const float *SFB = p.sfb + (int64_t)bg * p.sfb_gs;
    uint16_t *C = p.out + (int64_t)g * p.c_gs;

    // ---- per-thread LDS-DMA sources.  Chunk id c = it*NT + tid lands at LDS byte 16*c (wave-uniform base +
    //      16*lane); it holds source chunk (c&7) ^ x(row) of row c>>3.  A and B are read through buffer
    //      descriptors based at this tile's first row: the per-lane part is a 32-bit byte offset (row*ld + col), the
    //      k advance rides in the scalar offset, and a lane whose chunk lies beyond K is sent out of range, for
    //      which the hardware stores zeros (no branch, no zero page).
    //      DNT is a multiple of 64*4, so (c&7) and x(row) -- hence col -- do not depend on `it`.
    constexpr int DNT = Cfg::DNT;
    const int dtid = tid & (DNT - 1);
    constexpr uint32_t kOutOfRange = 0x80000000u;  // > num_records (host guarantees tile extents < 2^31)
    const int a_col = ((dtid & 7) ^ swz_a(dtid >> 3)) * 16;
    const int b_col = ((dtid & 7) ^ swz_b(dtid >> 3)) * 16;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::A_ITERS; ++it) {
        const int row = (it * DNT + dtid) >> 3;
        // rows at or beyond masked_m[g] (M): nothing is fetched for them -- the lane is sent out of the descriptor's range
        // and the hardware zero-fills its LDS bytes (their outputs are never stored); a dense tile's rows beyond M likewise
        const int rr = min(row, M - 1 - m0);
#ifdef DGA_ABL_IDX_NOA   // diagnostic (identity row table only): the A rows are addressed as in the packed layout
        a_voff[it] = row < M - m0 ? (uint32_t)(g * p.m + m0 + rr) * (uint32_t)p.lda + a_col : kOutOfRange;
#else
        a_voff[it] = row < M - m0 ? (ridx ? (uint32_t)ridx[m0 + rr] : (uint32_t)rr) * (uint32_t)p.lda + a_col : kOutOfRange;
#endif
        if constexpr (UNAL)   // byte offset from the dword at or below the matrix's first byte (rows start at any byte, and so may the matrix)
            a_voff[it] = row < M - m0 ? (uint32_t)(m0 + row) * (uint32_t)p.lda + a_col + (uint32_t)((uintptr_t)A & 3) : kOutOfRange;
    }
#pragma unroll
    for (int it = 0; it < Cfg::B_ITERS; ++it) {
        const int row = (it * DNT + dtid) >> 3;
        b_voff[it] = (uint32_t)min(row, p.n - 1 - n0) * (uint32_t)p.ldb + b_col;
        if constexpr (UNAL) b_voff[it] = (uint32_t)(n0 + min(row, p.n - 1 - n0)) * (uint32_t)p.ldb + b_col + (uint32_t)((uintptr_t)B & 3);
    }
    // (UNAL: from the dword at or below the matrix's first byte -- the per-lane offsets above count from there -- to the end of
    //  the dword that holds its last byte)
    const v4i a_rsrc = UNAL ? make_rsrc(A - ((uintptr_t)A & 3), ((int64_t)((uintptr_t)A & 3) + (int64_t)M * p.lda + 3) & ~(int64_t)3)
                            : (ridx ? make_rsrc(A, p.a_bytes) : make_rsrc(A + (int64_t)m0 * p.lda, (int64_t)(M - m0) * p.lda));
    const v4i b_rsrc = UNAL ? make_rsrc(B - ((uintptr_t)B & 3), ((int64_t)((uintptr_t)B & 3) + (int64_t)p.n * p.ldb + 3) & ~(int64_t)3)
                            : make_rsrc(B + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
    // scale slots: [0,BM) = sfa rows of this tile, [BM, BM+8) = sfb blocks of this tile, rest = padding
    const float *sc_src[Cfg::SC_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::SC_ITERS; ++it) {
        const int s = it * DNT + dtid;
        if (s < BM) {
            const int mr = min(m0 + s, M - 1);
#ifdef DGA_ABL_IDX_NOSC
            sc_src[it] = SFA + (int64_t)(ridx ? g * p.m + mr : mr) * p.sfa_ld;
#else
            sc_src[it] = SFA + (ridx ? ridx[mr] : (int64_t)mr) * p.sfa_ld;
#endif
        } else {
            const int nb = min(n0 / 128 + min(s - BM, 7), p.nb_n - 1);
            sc_src[it] = SFB + (int64_t)nb * p.kb_n;
        }
    }
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;

    // one LDS-DMA wave-instruction of a stage: idx in [0, LOADS_PER_STAGE) = A pieces, B pieces, scale pieces.
    // Inline asm on purpose: hipcc waits vmcnt(0) in front of every ds_read that follows an LDS-DMA builtin it can
    // see (it cannot tell the two stages apart), which would serialise the pipeline; the landing of these loads is
    // ordered by the loop's own vmcnt(0) + barrier instead.
    // `kb` past the last block (the refill issued from inside the last k block) is branch-free: every chunk is
    // then beyond K and zero-fills the idle stage; the scale gather re-reads the last block's scales.
    // One DMA = add (LDS address), s_mov m0, s_nop, buffer_load: every extra scalar or vector instruction here is paid
    // 9 times per wave per k block in issue slots the MFMA stream needs.  KTAIL (K % 128 != 0) adds the per-lane
    // beyond-K test; without it a refill that runs past the last k block (issued from inside the last blocks) just
    // reads the following bytes of the tile -- valid memory inside the descriptor's range -- into a stage nobody
    // consumes.
    auto issue_one = [&](int idx, int stage, int kb) {
        const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES + dwave * 1024;
        const int k0 = kb * 128;
        if (idx < Cfg::A_ITERS) {
#ifdef DGA_ABL_NOADMA
            return;   // diagnostic: the A tile is never fetched (whatever lies in the LDS is multiplied)
#endif
            const int it = idx;
            uint32_t voff = a_voff[it];
            if constexpr (KTAIL) voff = (k0 + a_col < p.k) ? voff : kOutOfRange;
            dma16(voff, a_rsrc, (uint32_t)k0, sa + it * DNT * 16);
        } else if (idx < Cfg::A_ITERS + Cfg::B_ITERS) {
            const int it = idx - Cfg::A_ITERS;
            uint32_t voff = b_voff[it];
            if constexpr (KTAIL) voff = (k0 + b_col < p.k) ? voff : kOutOfRange;
            // (the non-temporal policy on the bf16-exact builds' weight stream measured no difference: 1096 against 1107 us on
            //  256 x (128, 7168, 2048) -- that kernel is bound by its matrix rate there, profiles/r04_grouped_policy_perf.txt)
            dma16(voff, b_rsrc, (uint32_t)k0, sa + Cfg::A_BYTES + it * DNT * 16);
        } else {
            const int it = idx - Cfg::A_ITERS - Cfg::B_ITERS;
            dma4(sc_src[it] + min(kb, p.kb_n - 1), lds0 + stage * Cfg::STAGE_BYTES + Cfg::A_BYTES + Cfg::B_BYTES +
                                                       (it * DNT + dwave * 64) * 4);
        }
    };

    // ---- per-lane fragment read offsets (bytes inside a stage)
    const int li = lane & 15, kg = lane >> 4;
    // A-matrix tile (MFMA Bop): row = wm*(BM/WM) + 16*mt + li
    const int a_row = wm * (BM / Cfg::kWM) + li;
    const int a_off0 = a_row * 128 + ((kg ^ swz_a(a_row)) * 16);
    const int a_off1 = a_row * 128 + (((kg + 4) ^ swz_a(a_row)) * 16);
    // B-matrix tile (MFMA Aop): row = wn*(BN/WN) + 32*j + 8*(li>>2) + 4*h + (li&3)
    const int b_row = wn * (BN / WN) + 8 * (li >> 2) + (li & 3);
    const int b_off0 = Cfg::A_BYTES + b_row * 128 + ((kg ^ swz_b(b_row)) * 16);
    const int b_off1 = Cfg::A_BYTES + b_row * 128 + (((kg + 4) ^ swz_b(b_row)) * 16);
    const int sa_off = Cfg::A_BYTES + Cfg::B_BYTES + (wm * (BM / Cfg::kWM) + li) * 4;
    const int sb_off = Cfg::A_BYTES + Cfg::B_BYTES + (BM + (wn * (BN / WN)) / 128) * 4;

    // ---- epilogue (a lambda so that each wave-group path of the ping-pong loop ends in its own copy: no register
    //      assignment has to agree across the two paths): lane owns row m, columns n0w + 32*j + 8*(lane>>4) + [0,8)
    // indexed form: the destination rows of this lane's m-tiles, fetched now (needed only in the epilogue)
    int64_t out_row[TM];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = m0 + wm * (BM / Cfg::kWM) + li + mt * 16;
#ifdef DGA_ABL_IDX_NOOUT
        out_row[mt] = ridx ? (int64_t)g * p.m + m : (int64_t)m;
#else
        out_row[mt] = (ridx && m < M) ? ridx[m] : (int64_t)m;
#endif
    }
    auto epilogue = [&](v4f (&acc)[TM][TN]) {
        const int m_row = m0 + wm * (BM / Cfg::kWM) + li;
        const int n_base = n0 + wn * (BN / WN) + 8 * kg;
        if (p.splitk > 1) {
            // split-K: this block's fp32 partial tile goes to its slab; lane owns 4 consecutive n per MFMA tile
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
                        // (plain stores: written through -- as the bf16 rows of a short raster are, below -- the slabs leave the XCD's L2
                        //  and the combine launch reads them from the fabric: 128 x 4096 x 7168 19.7 -> 21.0 us, 128 x 7168 x 18432
                        //  42.8 -> 47.2; only slabs of a few MB gain, 64 x 2112 x 7168 13.9 -> 13.1.  scripts/r06_splitk_slab_ab.py)
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
            if (p.m_indices && p.m_indices[m] != bg) continue;  // a row of another group / a padding row: untouched
            uint16_t *crow = C + out_row[mt] * p.ldc;
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
#ifdef DGA_ABL_NOSTORE
                    if (lo.x == 12345.678f) *(v4i *)(crow + n) = pk;  // diagnostic: keep the math, drop the write stream
#else
                    // out_nt (dense one-tile-per-CU rasters): the tile's rows leave with a cache policy that does not park them in
                    // the XCD's L2 -- every CU stores its 128 KB at the same moment and the kernel-end write-back of what the L2
                    // still holds dirty would otherwise follow the store burst instead of running under it
                    // (inline asm: written as a builtin beside a plain store the branches are merged and the hint is lost)
                    if (p.out_nt == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(crow + n), "v"(pk) : "memory");
                    else if (p.out_nt == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(crow + n), "v"(pk) : "memory");
                    else if (p.out_nt == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(crow + n), "v"(pk) : "memory");
                    else *(v4i *)(crow + n) = pk;
#endif
                } else {
                    const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (n + q < p.n) crow[n + q] = e[q];
                }
            }
        }
    };

    if constexpr (PP == 1) {
        // ---- ping-pong main loop -----------------------------------------------------------------------------
        // Halves: X = waves 0-3 (A rows 0-127), Y = waves 4-7 (A rows 128-255); both read all of B.  A k block is two
        // half-phases of 16 MFMAs per wave: H1 = n-tiles 0-3 (B rows 0-63 of the wave's 128-row block), H2 = n-tiles
        // 4-7 (rows 64-127).  Two barriers per k block, Ba(kb) and Bb(kb), delimit intervals
        //     I1(kb) = Ba..Bb : X reads its fragments of kb and runs H1(kb)   |  Y runs H2(kb-1)
        //     I2(kb) = Bb..Ba': X runs H2(kb)                                 |  Y reads its fragments of kb, runs H1(kb)
        // so one half's fragment burst (the ~450 idle cycles per k block of the PP = 0 loop) hides under the other's
        // MFMAs.  The refill of a stage is split the same way:
        //     early(kb+1) = A tile, scales, B rows used by H1 -> issued during I1(kb), must have landed at Ba(kb+1)
        //     late(kb+1)  = B rows used by H2                 -> issued during I2(kb), must have landed at Bb(kb+1)
        // WAR: early(kb+1) overwrites what X last read in I1(kb-1) and Y in I2(kb-1) (both before Ba(kb));
        //      late(kb+1) overwrites what X last read in I2(kb-1) and Y in I1(kb) (both before Bb(kb)).
        // Each wave's DMA stream is E(0) L(0) | E(1) L(1) | ... in issue order, so "E(kb) landed" = all but the
        // youngest LATE_N, and "L(kb) landed" = all but the youngest EARLY_N (E(kb+1) is already in flight).
        static_assert(PP == 0 || (BM == 256 && BN == 256 && Cfg::kWM == 4 && WN == 2), "ping-pong: 256x256, 8 waves");
        constexpr int LAG = 3, RING = LAG + 1, HS = TM * TN / 2;
        constexpr int LATE_N = Cfg::B_ITERS / 2, EARLY_N = Cfg::A_ITERS + Cfg::B_ITERS / 2 + Cfg::SC_ITERS;
        static_assert(Cfg::B_ITERS * Cfg::DNT / 8 == BN && (Cfg::DNT / 8) * 2 == BN / WN, "B piece = half an n block");
        auto early_idx = [](int j) {
            return j < Cfg::A_ITERS ? j
                 : j < Cfg::A_ITERS + Cfg::B_ITERS / 2 ? Cfg::A_ITERS + 2 * (j - Cfg::A_ITERS)
                 : Cfg::A_ITERS + Cfg::B_ITERS + (j - Cfg::A_ITERS - Cfg::B_ITERS / 2);
        };
        auto late_idx = [](int j) { return Cfg::A_ITERS + 2 * j + 1; };
        v4f part[RING];
        v8i bf[2];
        v8i af[TM];
        float s[TM];
        float sfb_v = 0.f;
        auto read_frags = [&](const uint8_t *st) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const v4i lo = *(const v4i *)(st + a_off0 + mt * 2048);
                const v4i hi = *(const v4i *)(st + a_off1 + mt * 2048);
                af[mt] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
            sfb_v = *(const float *)(st + sb_off);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) s[mt] = *(const float *)(st + sa_off + mt * 64);
        };
        // one half-phase: H (0/1) selects the n-tiles, KIND the DMA list riding on it (0 none, 1 early, 2 late)
        auto run_half = [&](v4f (&acc)[TM][TN], const uint8_t *st, auto Hc, auto KINDc, auto FRESHc, int dstage, int dkb) {
            constexpr int H = decltype(Hc)::value, KIND = decltype(KINDc)::value;
            constexpr bool FRESH = decltype(FRESHc)::value;  // s[] holds raw sfa values: fold sfb in first
            constexpr int NDMA = KIND == 1 ? EARLY_N : (KIND == 2 ? LATE_N : 0);
            constexpr int ISSUE_STEPS = (HS * 3) / 4;
            {
                const int boff = ((H * 4) >> 1) * 4096;
                const v4i lo = *(const v4i *)(st + b_off0 + boff);
                const v4i hi = *(const v4i *)(st + b_off1 + boff);
                bf[0] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < HS + LAG; ++i) {
                if (i < HS) {
                    const int q = i / TM, mt = i % TM, nt = H * 4 + q;
                    part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        bf[q & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (NDMA > 0 && i < ISSUE_STEPS) {
#pragma unroll
                        for (int j = (i * NDMA) / ISSUE_STEPS; j < ((i + 1) * NDMA) / ISSUE_STEPS; ++j)
                            issue_one(KIND == 1 ? early_idx(j) : late_idx(j), dstage, dkb);
                    }
                    if (mt == 0 && q + 1 < 4) {
                        const int boff = ((nt + 1) >> 1) * 4096 + ((nt + 1) & 1) * 512;
                        const v4i lo = *(const v4i *)(st + b_off0 + boff);
                        const v4i hi = *(const v4i *)(st + b_off1 + boff);
                        bf[(q + 1) & 1] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
                }
                if (FRESH && i == LAG) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) s[mt] *= sfb_v;
                }
                if (i >= LAG) {
                    const int j = i - LAG, nt = H * 4 + j / TM, mt = j % TM;
                    const v4f pr = part[j % RING];
                    acc[mt][nt].x = __builtin_fmaf(pr.x, s[mt], acc[mt][nt].x);
                    acc[mt][nt].y = __builtin_fmaf(pr.y, s[mt], acc[mt][nt].y);
                    acc[mt][nt].z = __builtin_fmaf(pr.z, s[mt], acc[mt][nt].z);
                    acc[mt][nt].w = __builtin_fmaf(pr.w, s[mt], acc[mt][nt].w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using Bt = std::integral_constant<bool, true>;
        using Bf = std::integral_constant<bool, false>;
        const int KB = p.kb_n;
        DGA_STAMP_DECL
        DGA_STAMP_CLOCK(6, 7);
        // prologue: E(0) then L(0), in that order
#pragma unroll
        for (int j = 0; j < EARLY_N; ++j) issue_one(early_idx(j), 0, 0);
#pragma unroll
        for (int j = 0; j < LATE_N; ++j) issue_one(late_idx(j), 0, 0);
        auto barrier = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        if (wave < 4) {  // ---------------- X
            v4f acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
            DGA_STAMP_START();
            for (int kb = 0; kb < KB; ++kb) {
                const uint8_t *st = smem + (kb & 1) * Cfg::STAGE_BYTES;
                wait_vmcnt<LATE_N>();
                barrier();  // Ba(kb)
                DGA_STAMP(0);
                read_frags(st);
                DGA_STAMP(1);
                run_half(acc, st, I0{}, I1{}, Bt{}, (kb & 1) ^ 1, kb + 1);
                DGA_STAMP(2);
                wait_vmcnt<EARLY_N>();
                barrier();  // Bb(kb)
                DGA_STAMP(3);
                run_half(acc, st, I1{}, I2{}, Bf{}, (kb & 1) ^ 1, kb + 1);
                DGA_STAMP(4);
            }
            barrier();  // Ba(KB): Y starts its last H2 while X stores
            DGA_STAMP_CLOCK(6, 7);
            DGA_STAMP_FLUSH();
            epilogue(acc);
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();  // Bb(KB): matches Y's last barrier
        } else {         // ---------------- Y
            v4f acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
            // first and last interval pairs are peeled so that the steady-state loop has no conditional around a
            // half-phase (a conditionally reloaded A fragment keeps both copies alive: +32 VGPRs, spills)
            wait_vmcnt<LATE_N>();
            barrier();  // Ba(0): nothing to compute yet, just start the refill
#pragma unroll
            for (int j = 0; j < EARLY_N; ++j) issue_one(early_idx(j), 1, 1);
            wait_vmcnt<EARLY_N>();
            barrier();  // Bb(0)
            read_frags(smem);
            run_half(acc, smem, I0{}, I2{}, Bt{}, 1, 1);
            DGA_STAMP_START();
            for (int kb = 1; kb < KB; ++kb) {
                const uint8_t *st = smem + (kb & 1) * Cfg::STAGE_BYTES;
                wait_vmcnt<LATE_N>();
                barrier();  // Ba(kb)
                DGA_STAMP(0);
                run_half(acc, smem + ((kb - 1) & 1) * Cfg::STAGE_BYTES, I1{}, I1{}, Bf{}, (kb & 1) ^ 1, kb + 1);
                DGA_STAMP(4);
                wait_vmcnt<EARLY_N>();
                barrier();  // Bb(kb)
                DGA_STAMP(3);
                read_frags(st);
                DGA_STAMP(1);
                run_half(acc, st, I0{}, I2{}, Bt{}, (kb & 1) ^ 1, kb + 1);
                DGA_STAMP(2);
            }
            wait_vmcnt<0>();
            barrier();  // Ba(KB)
            run_half(acc, smem + ((KB - 1) & 1) * Cfg::STAGE_BYTES, I1{}, I0{}, Bf{}, 0, 0);
            barrier();  // Bb(KB)
            DGA_STAMP_CLOCK(6, 7);
            DGA_STAMP_FLUSH();
            epilogue(acc);
        }
    } else if constexpr (PP == 2) {
        // ---- continuous pipeline ---------------------------------------------------------------------------------
        // The k-block boundary disappears from the MFMA stream.  Per k block (stage s = kb & 1), steps i = 0..STEPS-1:
        //   * every step: MFMA(i) -> part ring; FMAs of step i-LAG (the first LAG steps promote the LAST LAG results of
        //     the previous k block, with that block's scales);
        //   * B fragments rotate through two register sets, the rotation simply continues into the next k block;
        //   * step SB (after the last LDS read of stage s has returned): vmcnt(0) + the ONE barrier.  Passing it means
        //     stage s^1 (k block kb+1, whose DMA was issued during the steps since the previous barrier) has landed
        //     everywhere AND nobody reads stage s any more;
        //   * steps STEPS-TM..STEPS-1: as each A fragment sees its last MFMA it is reloaded IN PLACE from stage s^1,
        //     together with the next block's first B fragment and scales -- their LDS latency hides under the block's
        //     last MFMAs and the next block's first step finds its operands in registers;
        //   * the refill of stage s (k block kb+2) starts right behind the barrier and continues over the first steps
        //     of the next k block: every batch has more than half a k block to land.
        constexpr int STEPS = TM * TN, LAG = 3, RING = 4;
        static_assert(STEPS % RING == 0 && STEPS > 2 * TM + LAG, "ring positions must line up across k blocks");
        constexpr int SB = STEPS - TM - 1;         // barrier step (just before the in-place reloads)
        constexpr int NL = Cfg::LOADS_PER_STAGE;
        // Placement of the refill DMA (measured, stamped build, cycles per 32 k blocks at 4096^3): spread thin over 18
        // steps 91.3k; 4 behind the barrier + 5 on the first 9 steps 83.8k.  The DMA costs little to ISSUE (an
        // all-lanes-out-of-range build runs as fast as one with no DMA at all); what costs is data that has not landed
        // when the vmcnt(0) at step SB asks for it -- so issue as early as the stage is free.
#ifndef DGA_TAIL_DMA
#define DGA_TAIL_DMA (TM < NL / 2 ? TM : NL / 2)
#endif
#ifndef DGA_HEAD_STEPS
#define DGA_HEAD_STEPS ((STEPS * 9) / 32 > NL - DGA_TAIL_DMA ? (STEPS * 9) / 32 : NL - DGA_TAIL_DMA)
#endif
        constexpr int TAIL_DMA = DGA_TAIL_DMA < NL ? DGA_TAIL_DMA : NL - 1;  // DMA slots on the steps behind the barrier
        constexpr int HEAD_STEPS = DGA_HEAD_STEPS;  // ... the rest rides on the first steps of the next k block
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
        // MATH = 2: the block's scales as E8M0 bytes (the fp32 exponent field) in byte 0 of a register: sae[mt] for this lane's
        // row of m-tile mt (the MFMA's second operand), sbe for the wave's 128-column block (first operand)
        int sae[TM], sbe = 0;
        auto e8m0 = [](float v) { return (int)((uint32_t)__builtin_bit_cast(int, v) >> 23); };
        const int KB = p.kb_n;
        DGA_STAMP_DECL
        DGA_STAMP_ABS(0);        // (diagnostic builds: wave entry, a few set-up instructions late)
        DGA_STAMP_CLOCK(6, 7);
        loop_clock.tick();
        auto barrier = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
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
        // DMA schedule.  Block d's refill = tail part (slots 0..TAIL_DMA-1), issued on the steps behind the barrier of
        // k block d-2, then head part (the rest), issued on the first HEAD_STEPS steps of k block d-1; the vmcnt(0) at
        // step SB of k block d-1 covers both.  Prologue: all of block 0, then tail(1).
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_one(idx, 0, 0);
#pragma unroll
        for (int idx = 0; idx < TAIL_DMA; ++idx) issue_one(idx, 1, 1);
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
                if constexpr (MATH == 2) sae[mt] = e8m0(*(const float *)(smem + sa_off + mt * 64));
                s_prev[mt] = 0.f;   // the first LAG "previous block" promotions add part (= 0) * 0
                s_next[mt] = 0.f;
            }
            if constexpr (MATH == 2) sbe = e8m0(sfb0);
        }
        DGA_STAMP_ABS(1);        // first fragments in registers: the k loop starts
        for (int kb = 0; kb < KB; ++kb) {
            const uint8_t *st = smem + (kb & 1) * Cfg::STAGE_BYTES;
            const uint8_t *sn = smem + ((kb & 1) ^ 1) * Cfg::STAGE_BYTES;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                const int nt = i / TM, mt = i % TM;
                if (i == SB) {
#ifndef DGA_ABL_NOWAIT   // diagnostic (results are garbage): the refill is issued and moves its bytes, nobody waits for it --
                         // the bound of what a deeper ring could save (a ring removes the WAIT, not the traffic)
                    wait_vmcnt<0>();
#endif
                    barrier();
                }
                if constexpr (MATH == 2)
                    acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[nt & 1], af[mt], acc[mt][nt], 0, 0, 0, sbe, 0, sae[mt]);
                else
                part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    bf[nt & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#ifndef DGA_ABL_NODMA
                // refill DMA: head part of block kb+1 (stage s^1 is free since the previous barrier) on the first
                // HEAD_STEPS steps, tail part of block kb+2 into THIS stage on the TM steps behind the barrier.
                // (Giving each SIMD its own issue steps -- 4 predicated copies of every slot -- measured 13 % MORE
                // cycles: the skipped copies cost more issue slots than the queueing they avoid.)
                if (i < HEAD_STEPS) {
#pragma unroll
                    for (int j = (i * (NL - TAIL_DMA)) / HEAD_STEPS; j < ((i + 1) * (NL - TAIL_DMA)) / HEAD_STEPS; ++j)
                        issue_one(TAIL_DMA + j, (kb & 1) ^ 1, kb + 1);
                }
                if (i > SB) {
#pragma unroll
                    for (int j = ((i - SB - 1) * TAIL_DMA) / TM; j < ((i - SB) * TAIL_DMA) / TM; ++j)
                        issue_one(j, kb & 1, kb + 2);
                }
#endif
                // B fragment rotation (continues into the next k block at the last n-tile)
#ifndef DGA_ABL_NOLDS
                if (mt == 0) {
                    if (nt + 1 < TN) bf[(nt + 1) & 1] = read_b(st, nt + 1);
                    else bf[(nt + 1) & 1] = read_b(sn, 0);
                }
#else
                if (mt == 0) { bf[(nt + 1) & 1] = bf[nt & 1]; asm volatile("" : "+v"(bf[(nt + 1) & 1])); }
#endif
                // in-place reload of the A fragment that has just seen its last MFMA of this k block
#ifdef DGA_ABL_NOLDS
                if (false) {
#else
                if (nt == TN - 1) {
#endif
                    af[mt] = read_a(sn, mt);
                    if (mt == 0) sfb_next = *(const float *)(sn + sb_off);
                    s_next[mt] = *(const float *)(sn + sa_off + mt * 64);
                }
                // promotion of step i - LAG
                if constexpr (MATH == 2) {
                    // (nothing: the MFMA accumulated in place)
                } else
                if (i >= LAG) {
                    const int j = i - LAG, jn = j / TM, jm = j % TM;
                    const v4f pr = part[j % RING];
#ifdef DGA_ABL_NOFMA
                    asm volatile("" :: "v"(pr), "v"(s[jm]));
#else
                    acc[jm][jn].x = __builtin_fmaf(pr.x, s[jm], acc[jm][jn].x);
                    acc[jm][jn].y = __builtin_fmaf(pr.y, s[jm], acc[jm][jn].y);
                    acc[jm][jn].z = __builtin_fmaf(pr.z, s[jm], acc[jm][jn].z);
                    acc[jm][jn].w = __builtin_fmaf(pr.w, s[jm], acc[jm][jn].w);
#endif
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
                if constexpr (MATH == 2) {
                    sae[mt] = e8m0(s_next[mt]);
                } else {
                    s_prev[mt] = s[mt];
                    s[mt] = s_next[mt] * sfb_next;
                }
            }
            if constexpr (MATH == 2) sbe = e8m0(sfb_next);
        }
        // drain: the last LAG results of the last k block
#pragma unroll
        for (int i = 0; i < (MATH == 2 ? 0 : LAG); ++i) {
            const int j = STEPS - LAG + i, jn = j / TM, jm = j % TM;
            const v4f pr = part[j % RING];
            acc[jm][jn].x = __builtin_fmaf(pr.x, s_prev[jm], acc[jm][jn].x);
            acc[jm][jn].y = __builtin_fmaf(pr.y, s_prev[jm], acc[jm][jn].y);
            acc[jm][jn].z = __builtin_fmaf(pr.z, s_prev[jm], acc[jm][jn].z);
            acc[jm][jn].w = __builtin_fmaf(pr.w, s_prev[jm], acc[jm][jn].w);
        }
        wait_vmcnt<0>();
        DGA_STAMP_CLOCK(6, 7);
        DGA_STAMP_ABS(2);
        loop_clock.tick();
        loop_clock.flush(p.stamps, blockIdx.x * (NT / 64) + wave, lane);
        epilogue(acc);
        DGA_STAMP_ABS(3);        // stores issued (not yet complete)
        DGA_STAMP_FLUSH();
    } else if constexpr (MATH == 1 || MATH == 3) {
        // ---- bf16-exact main loop (dispatchPolicyTag 7) ---------------------------------------------------------------
        // Numerics: every e4m3 value is a bf16 value, so v_cvt_scalef32_pk_bf16_fp8 (scale 1) converts exactly; the products
        // are exact in fp32 and v_mfma_f32_16x16x32_bf16 sums them with fp32-class error (profiles/r02_mfma_forms.txt: 2^-25 S
        // on amax-quantised data, 99.75 % of the 128-wide block sums bit-equal to the oracle's), against the 2^-16 S of the fp8
        // forms.  One scale block = four chained MFMAs (C = 0, then C = the chain), then the same fp32 promotion as the fast
        // path.  The LDS image, the DMA and the fragment reads are the fast path's: lane (li, kg) holds bytes [16 kg, +16) and
        // [64 + 16 kg, +16) of its row; MFMA q of the chain takes dwords 2q, 2q + 1 of those 32 bytes from BOTH operands.
        // Cost model (scripts/ubench/bf16x_ubench.hip): the loop is bound by vector issue, not by the matrix pipe -- per
        // MFMA one promotion FMA plus the conversions (A fragments are re-converted by the WN waves that share them, B by
        // the WM waves).  The accumulators, the bf16 A fragments (16 registers per m-tile) and a double-buffered bf16 B
        // fragment have to fit 256 registers at two waves per SIMD: wave tiles of at most 64 x 64.
        // Schedule: three LDS stages = {being consumed, landed and readable, being filled}.  ONE barrier per k block, at its
        // top, behind vmcnt(0): block kb + 1 has landed everywhere, and the stage block kb + 2 goes to is free.  Inside a
        // block the tiles run (n-tile outer, m-tile inner); a "gap" is the issue slot behind one MFMA:
        //   * B(nt + 1) is converted during n-tile nt into the other bf16 set (the last n-tile converts the NEXT block's
        //     B(0) from the readable stage); its raw bytes sit in one 8-register buffer whose halves are reloaded as the
        //     conversions release them;
        //   * A fragments are re-converted IN PLACE for the next block as they die: A[mt] sees its last MFMA in tile
        //     (mt, TN - 1) and is converted during the following tile's four gaps (A[TM - 1] during the next block's first
        //     tile), raw bytes read one tile ahead into two alternating buffers;
        //   * the promotion of tile t rides on tile t + LAGT (the first tiles of a block promote the previous block's last
        //     ones with that block's scales); the refill DMA rides on the second tile onward.
        // Measured (scripts/ubench/stamp_tile_bx128x256, profiles/r03_bf16_exact.txt): 3440 ticks per k block against 2048 of
        // matrix pipe and ~2870 of the register-resident loop with the same vector work (dga_mfma_ceiling mode 1): the gap is the
        // barrier (the older wave of each SIMD wins every arbitration, finishes its block in 2036 ticks and waits 1222 for its
        // partner; a priority that falls as a wave advances -- s_setprio 3..0 per quarter block -- evens the two out, 2699 / 3119,
        // and changes the total by 0.8 %: the sum of the two waves' issue slots is what counts, not who takes them).
        static_assert(PP == 0 && Cfg::STAGES == 3 && !LC, "bf16-exact: plain loop, three stages, no loader waves");
        static_assert((TM == 2 || TM == 4) && (TN == 2 || TN == 4 || (MATH == 3 && TN == 8)), "bf16-exact: wave tiles of 32..64 x 32..64 (x 128 for MATH = 3)");
        constexpr bool UE = MATH == 3;    // power-of-two scales folded into the A conversions, the MFMA chain accumulates in place
        constexpr int STG = 3, NL = Cfg::LOADS_PER_STAGE, TILES = TM * TN, G = 4 * TM, LAGT = 2, RING = 4;
        static_assert(TILES % RING == 0 && TILES > LAGT && 4 * TILES >= 4 + NL && 16 % G == 0, "ring positions / DMA slots line up");
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        v4f acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        v4f part[RING];
#pragma unroll
        for (int i = 0; i < RING; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
        v4i afx[TM][4], bfx[2][4];      // bf16 fragments: [q] = the 8 bf16 of MFMA q of the chain
        v4i braw[2], araw[2][2];         // raw e4m3 bytes: [0] = bytes [16 kg, +16), [1] = bytes [64 + 16 kg, +16)
        float s_cur[TM], s_old[TM], s_nxt[TM];
        // conversion c (0..15) of a fragment: dword c >> 1 of the 32 raw bytes, half c & 1 -> dword c & 3 of MFMA (c >> 2).
        // (c is a constant after unrolling; the builtin's half selector must be an immediate)
        auto convert = [](const v4i (&raw)[2], v4i (&dst)[4], int c, float scale = 1.0f) {
#ifdef DGA_ABL_BX_NOCVT   // diagnostic (results are garbage): the MFMAs run on whatever the fragment registers hold
            asm volatile("" : "+v"(dst[c >> 2][c & 3]));
            return;
#endif
            const int w = raw[(c >> 1) >> 2][(c >> 1) & 3];
            dst[c >> 2][c & 3] = (c & 1) ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, scale, true))
                                         : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, scale, false));
        };
        auto b_frag_off = [](int nt) { return (nt >> 1) * 4096 + (nt & 1) * 512; };
        // prologue: blocks 0 and 1 on their way, block 0 landed; its fragments converted in one burst (once per tile)
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int idx = 0; idx < NL; ++idx) issue_one(idx, d, kb_begin + d);
        wait_vmcnt<NL>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {
            const float sfb0 = *(const float *)(smem + sb_off);
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                // (buffer mt & 1: block 0's first tile converts A[TM - 1] once more from the buffer it was read into)
                araw[mt & 1][0] = *(const v4i *)(smem + a_off0 + mt * 2048);
                araw[mt & 1][1] = *(const v4i *)(smem + a_off1 + mt * 2048);
                s_cur[mt] = *(const float *)(smem + sa_off + mt * 64) * sfb0;
#pragma unroll
                for (int c = 0; c < 16; ++c) convert(araw[mt & 1], afx[mt], c, UE ? s_cur[mt] : 1.0f);
                s_old[mt] = 0.f;    // the first LAGT tiles "promote the previous block": part (= 0) * 0
                s_nxt[mt] = 0.f;
            }
            braw[0] = *(const v4i *)(smem + b_off0);
            braw[1] = *(const v4i *)(smem + b_off1);
#pragma unroll
            for (int c = 0; c < 16; ++c) convert(braw, bfx[0], c);
            braw[0] = *(const v4i *)(smem + b_off0 + b_frag_off(1));   // B(1) of block 0, raw
            braw[1] = *(const v4i *)(smem + b_off1 + b_frag_off(1));
        }
        DGA_STAMP_DECL
        DGA_STAMP_START();
        loop_clock.tick();
        int cur = 0, nxt = 1, fill = 2;
        // A wave whose rows all lie at or beyond M -- the masked grouped layout with an expert that has fewer rows than the tile is
        // tall, a dense problem's ragged last tile row -- multiplies nothing: it keeps its share of the refill DMA and the barriers,
        // and its SIMD partner gets the issue slots (256 x (128, 7168, 2048), random masks, two runs each on one box: 888 / 904 -> 853 / 870 us;
        // full mask unchanged, 983 -> 988).
        const bool rows_present = m0 + wm * (BM / Cfg::kWM) < M;   // (wave-uniform)
        if (!rows_present) {
            for (int kb = kb_begin; kb < kb_end; ++kb) {
                wait_vmcnt<0>();
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
#pragma unroll
                for (int idx = 0; idx < NL; ++idx) issue_one(idx, fill, kb + 2);
                const int f = cur;
                cur = nxt; nxt = fill; fill = f;
            }
            wait_vmcnt<0>();
            return;
        }
        for (int kb = kb_begin; kb < kb_end; ++kb) {
            wait_vmcnt<0>();                         // this wave's pieces of block kb + 1 (issued a block ago) have landed
            DGA_STAMP(1);
            __builtin_amdgcn_s_barrier();            // ... everyone's have; and everyone has left block kb - 1, whose stage
            asm volatile("" ::: "memory");           //     block kb + 2 is about to overwrite
            DGA_STAMP(2);
            const uint8_t *sc = smem + cur * Cfg::STAGE_BYTES;   // being consumed (B raw reloads of this block)
            const uint8_t *sn = smem + nxt * Cfg::STAGE_BYTES;   // landed: the next block's fragments are read ahead from it
#pragma unroll
            for (int u = 0; u < 4 * TILES; ++u) {
                const int t = u >> 2, q = u & 3, nt = t / TM, mt = t % TM, g = u % G;
                if constexpr (UE)    // the scales are inside the A fragments: the chain runs on through every block
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        __builtin_bit_cast(v8bf, bfx[nt & 1][q]), __builtin_bit_cast(v8bf, afx[mt][q]), acc[mt][nt], 0, 0, 0);
                else
                part[t % RING] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(v8bf, bfx[nt & 1][q]), __builtin_bit_cast(v8bf, afx[mt][q]),
                    q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t % RING], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // refill DMA of block kb + 2: one piece per gap from the second tile on
#ifndef DGA_ABL_BX_NODMA  // diagnostic (results are garbage): no refill -- the stages keep the prologue's bytes
                if (u >= 4 && u < 4 + NL) issue_one(u - 4, fill, kb + 2);
#endif
                // B(nt + 1) -> bfx[(nt + 1) & 1]: 16 conversions over the n-tile's G gaps; the raw halves are reloaded with
                // B(nt + 2) as they are released (from the readable stage once nt + 2 runs past this block)
#pragma unroll
                for (int c = 0; c < 16 / G; ++c) convert(braw, bfx[(nt + 1) & 1], (16 / G) * g + c);
                {
                    const int nn = nt + 2;
                    const uint8_t *src = nn < TN ? sc : sn;
                    const int off = b_frag_off(nn < TN ? nn : nn - TN);
#ifndef DGA_ABL_BX_NOLDS  // diagnostic (results are garbage): no fragment reads inside the loop
                    if (g == G / 2 - 1) braw[0] = *(const v4i *)(src + b_off0 + off);
                    if (g == G - 1) braw[1] = *(const v4i *)(src + b_off1 + off);
#endif
                }
                // A fragments of the NEXT block, in place: raw bytes of A[mt] one tile ahead (at the first gap of its last
                // tile), conversion during the tile after its last one
#ifndef DGA_ABL_BX_NOLDS
                if (nt == TN - 1 && q == 0) {
                    araw[mt & 1][0] = *(const v4i *)(sn + a_off0 + mt * 2048);
                    araw[mt & 1][1] = *(const v4i *)(sn + a_off1 + mt * 2048);
                }
#endif
                if (nt == TN - 1 && mt >= 1) {     // the tile behind (mt - 1, TN - 1): A[mt - 1] of the next block
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(mt - 1) & 1], afx[mt - 1], 4 * q + c, UE ? s_nxt[mt - 1] : 1.0f);
                }
                if (t == 0) {                      // the tile behind the previous block's (TM - 1, TN - 1): A[TM - 1] of THIS block
#pragma unroll
                    for (int c = 0; c < 4; ++c) convert(araw[(TM - 1) & 1], afx[TM - 1], 4 * q + c, UE ? s_cur[TM - 1] : 1.0f);
                }
                // the next block's scales: needed from its first promotions, LAGT tiles into it -- or (UE) by the conversions of its A
                // fragments, which start in this block's last n-tile: read at the block's first gap (the stage has landed)
                if (u == (UE ? 0 : 4 * TILES - 8)) {
                    const float sfbn = *(const float *)(sn + sb_off);
#pragma unroll
                    for (int i = 0; i < TM; ++i) s_nxt[i] = *(const float *)(sn + sa_off + i * 64) * sfbn;
                }
                // promotion of tile t - LAGT, one accumulator element per gap
                if constexpr (!UE) {
                    const int j = t >= LAGT ? t - LAGT : TILES + t - LAGT, jn = j / TM, jm = j % TM;
                    const float sv = t >= LAGT ? s_cur[jm] : s_old[jm];
#ifdef DGA_ABL_BX_NOFMA   // diagnostic (results are garbage): no promotion
                    asm volatile("" :: "v"(part[j % RING][q]), "v"(sv));
#else
                    acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], sv, acc[jm][jn][q]);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            DGA_STAMP(4);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                s_old[i] = s_cur[i];
                s_cur[i] = s_nxt[i];
            }
            const int f = cur;
            cur = nxt; nxt = fill; fill = f;
        }
        // drain: the last LAGT tiles of the last block
        if constexpr (!UE) {
#pragma unroll
        for (int t = 0; t < LAGT; ++t) {
            const int j = TILES + t - LAGT, jn = j / TM, jm = j % TM;
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[jm][jn][q] = __builtin_fmaf(part[j % RING][q], s_old[jm], acc[jm][jn][q]);
        }
        }
        wait_vmcnt<0>();   // the refills past the last k block land in LDS nobody reads: drain them before the stores / exit
        DGA_STAMP_CLOCK(6, 7);
        DGA_STAMP_FLUSH();
        loop_clock.tick();
        loop_clock.flush(p.stamps, blockIdx.x * (NT / 64) + wave, lane);
        epilogue(acc);
    } else {
        v4f acc[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // ---- plain main loop (PP = 0).  ONE barrier per k block: passing it means (a) every wave's DMA of this stage
        // has landed (each waited vmcnt first) and (b) every wave has left the previous k block, i.e. the stage being
        // refilled is free -- so its refill is issued from inside this k block's MFMA pipeline, one DMA wave-instruction
        // every few MFMAs.  (Issued in a burst at the top, the 9 DMA instructions cost a wave 800-1600 cycles of blocked
        // issue per k block; measured with the -DDGA_STAMPS build.)
        // With STAGES = 3 (tiles whose stage is <= 48 KB) two refills are in flight instead of one: the wait at the
        // top leaves the newest batch outstanding.
        constexpr int STG = Cfg::STAGES;
        DGA_STAMP_DECL
        // first fills: the B and scale pieces go out before the A pieces, so that in the indexed form the row-table loads
        // behind a_voff have the B issue to hide under (per stage the piece count is what the vmcnt waits rely on,
        // not the order inside a stage)
        if ((!LC || loader) && !UNAL) {
#pragma unroll
            for (int d = 0; d < STG - 1; ++d) {
#pragma unroll
                for (int idx = Cfg::A_ITERS; idx < Cfg::LOADS_PER_STAGE; ++idx) issue_one(idx, d, kb_begin + d);
#pragma unroll
                for (int idx = 0; idx < Cfg::A_ITERS; ++idx) issue_one(idx, d, kb_begin + d);
            }
        }
        DGA_STAMP_START();
        DGA_STAMP_CLOCK(6, 7);   // slots 6/7: shader-clock and 100 MHz real-time ticks across the main loop
        loop_clock.tick();
        int stage = 0, fill = STG - 1;   // stage being consumed / stage being refilled (with k block kb + STG - 1)
        const bool wave_has_rows = m0 + wm * (BM / Cfg::kWM) < M;  // wave-uniform (wm comes from readfirstlane)
        if constexpr (LC && UNAL) {
            if (loader) {
                // ---- loader wave, rows that start at any byte.  The LDS-DMA (and a plain buffer load) takes a source address
                // of ANY alignment at full rate on this part (scripts/ubench/probe_unaligned_dma.hip ->
                // profiles/r04_probe_unaligned_dma.txt: 16 misalignments exact, 5.7-6.0 TB/s each), so every k block but one
                // is fetched exactly as in the aligned build, with byte offsets from the matrix's first byte.  The one block
                // that holds K's ragged end (K % 16 != 0: the chunk that straddles K would bring the NEXT row's first bytes
                // into the image) goes through registers: buffer_load_dwordx4, the bytes at and beyond K zeroed, ds_write_b128
                // to the address the DMA would have written.  The scales always go by LDS-DMA.
                constexpr int NP = Cfg::A_ITERS + Cfg::B_ITERS;
                const int64_t a_mis = (int64_t)((uintptr_t)A & 3), b_mis = (int64_t)((uintptr_t)B & 3);
                const __amdgpu_buffer_rsrc_t ua = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(A - a_mis), 0, (int)min((a_mis + (int64_t)M * p.lda + 3) & ~(int64_t)3, (int64_t)0x7FFFFFFC), 0x00020000);
                const __amdgpu_buffer_rsrc_t ub = __builtin_amdgcn_make_buffer_rsrc(
                    (void *)(B - b_mis), 0, (int)min((b_mis + (int64_t)p.n * p.ldb + 3) & ~(int64_t)3, (int64_t)0x7FFFFFFC), 0x00020000);
                const int ragged_kb = (p.k & 15) ? p.k / 128 : -1;    // the k block that holds the straddling chunk
                bool drained = false;                                 // the ragged block is behind us: no batch counting any more
                auto fill_stage = [&](int stg, int kbr) {
                    if (kbr != ragged_kb) {
#pragma unroll
                        for (int idx = Cfg::A_ITERS; idx < Cfg::LOADS_PER_STAGE; ++idx) issue_one(idx, stg, kbr);
#pragma unroll
                        for (int idx = 0; idx < Cfg::A_ITERS; ++idx) issue_one(idx, stg, kbr);
                        return;
                    }
                    // five dwords from the dword-aligned address at or below the chunk (each load a whole aligned dword, range-checked
                    // on its own: a 16-byte load that ends past num_records comes back as zeros altogether -- measured), shifted into
                    // place by v_alignbyte
                    const int k0 = kbr * 128;
                    uint32_t r[NP][5];
#pragma unroll
                    for (int idx = 0; idx < NP; ++idx) {
                        const bool isa = idx < Cfg::A_ITERS;
                        const uint32_t vo = isa ? a_voff[idx] : b_voff[idx - Cfg::A_ITERS];
                        const bool live = vo != kOutOfRange && k0 + (isa ? a_col : b_col) < p.k;
                        const uint32_t al = live ? (vo & ~3u) + (uint32_t)k0 : kOutOfRange;
                        // (builtin loads, not asm: the compiler must know when these registers are valid -- with asm loads it is free
                        //  to copy an output register before the hand-placed wait, and did: one run in three came out wrong)
#pragma unroll
                        for (int j = 0; j < 5; ++j) r[idx][j] = __builtin_amdgcn_raw_buffer_load_b32(isa ? ua : ub, (int)(al + 4 * j), 0, 0);
                    }
#pragma unroll
                    for (int idx = 0; idx < NP; ++idx) {
                        const bool isa = idx < Cfg::A_ITERS;
                        const uint32_t sh = (isa ? a_voff[idx] : b_voff[idx - Cfg::A_ITERS]) & 3u;
                        const int valid = p.k - (k0 + (isa ? a_col : b_col));   // bytes of this chunk inside the row
                        v4i w;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            uint32_t x = __builtin_amdgcn_alignbyte(r[idx][j + 1], r[idx][j], sh);
                            const int nb = valid - 4 * j;
                            if (nb < 4) x = nb <= 0 ? 0u : (x & ((1u << (8 * nb)) - 1u));
                            w[j] = (int)x;
                        }
                        const int it = isa ? idx : idx - Cfg::A_ITERS;
                        *(v4i *)(smem + stg * Cfg::STAGE_BYTES + (isa ? 0 : Cfg::A_BYTES) + dwave * 1024 + it * DNT * 16 + lane * 16) = w;
                    }
#pragma unroll
                    for (int idx = NP; idx < Cfg::LOADS_PER_STAGE; ++idx) issue_one(idx, stg, kbr);   // its scales
                    drained = true;
                };
#pragma unroll
                for (int d = 0; d < STG - 1; ++d) fill_stage(d, kb_begin + d);
                for (int kb = kb_begin; kb < kb_end; ++kb) {
                    if (drained) {
                        wait_vmcnt<0>();
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the ragged block's image stores
                    } else {
                        wait_vmcnt<(STG - 2) * Cfg::LOADS_PER_STAGE>();
                    }
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                    fill_stage(fill, kb + STG - 1);
                    fill = fill + 1 == STG ? 0 : fill + 1;
                }
                wait_vmcnt<0>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                return;
            }
        } else if constexpr (LC) {
            if (loader) {
                // loader wave: per k block, wait until the oldest batch in flight has landed, meet the computing waves at
                // the barrier (they have left the stage that is refilled next), send the whole refill in one burst
                for (int kb = kb_begin; kb < kb_end; ++kb) {
                    wait_vmcnt<(STG - 2) * Cfg::LOADS_PER_STAGE>();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
#ifndef DGA_ABL_NODMA
#pragma unroll
                    for (int idx = Cfg::A_ITERS; idx < Cfg::LOADS_PER_STAGE; ++idx) issue_one(idx, fill, kb + STG - 1);
#pragma unroll
                    for (int idx = 0; idx < Cfg::A_ITERS; ++idx) issue_one(idx, fill, kb + STG - 1);
#endif
                    fill = fill + 1 == STG ? 0 : fill + 1;
                }
                wait_vmcnt<0>();   // the refills past the last k block land in LDS nobody reads: drain them before exit
                return;
            }
        }
        for (int kb = kb_begin; kb < kb_end; ++kb) {
            if constexpr (!LC) wait_vmcnt<(STG - 2) * Cfg::LOADS_PER_STAGE>();
            DGA_STAMP(1);                            // segment 1: vmcnt wait
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");           // no LDS read may be hoisted above the barrier
            DGA_STAMP(2);                            // segment 2: barrier "stage ready, other stage free"

            const uint8_t *st = smem + stage * Cfg::STAGE_BYTES;
            if (!wave_has_rows) {
                // masked-M: every row of this wave's m range is >= masked_m[g].  It still carries its share of the
                // refill (and the barrier), but computes nothing.
                if constexpr (!LC) {
#pragma unroll
                    for (int idx = 0; idx < Cfg::LOADS_PER_STAGE; ++idx) issue_one(idx, fill, kb + STG - 1);
                }
                stage = stage + 1 == STG ? 0 : stage + 1;
                fill = fill + 1 == STG ? 0 : fill + 1;
                continue;
            }
            // Fragment reads are ordered so that the first MFMA waits only for ITS operands (B n-tile 0, A m-tile 0):
            // all 8 waves hit the LDS at once here, and a wave that waited for its whole 15-read burst would idle the
            // matrix pipe for ~450 cycles per k block.  The scale reads come last; they are first needed LAG steps later.
            constexpr int STEPS = TM * TN, LAG = 3, RING = LAG + 1;
#ifndef DGA_ISSUE_STEPS
#define DGA_ISSUE_STEPS ((STEPS * 5) / 8)
#endif
            constexpr int ISSUE_STEPS = DGA_ISSUE_STEPS > 0 ? (DGA_ISSUE_STEPS < STEPS ? DGA_ISSUE_STEPS : STEPS) : 1;  // refill DMA rides on the first steps
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
            // MATH = 2: the scales as E8M0 bytes (the fp32 exponent field) in byte 0 of a register
            int sae[TM], sbe = 0;
            if constexpr (MATH == 2) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) sae[mt] = (int)((uint32_t)__builtin_bit_cast(int, s[mt]) >> 23);
                sbe = (int)((uint32_t)__builtin_bit_cast(int, sfb_v) >> 23);
            }
            DGA_STAMP(3);                            // segment 3: first fragments + scales out of LDS
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < STEPS + LAG; ++i) {
                if (i < STEPS) {
                    const int nt = i / TM, mt = i % TM;
#ifdef DGA_ABL_NOMFMA
                    part[i % RING] = v4f{0.f, 0.f, 0.f, 0.f};   // diagnostic: the stream without the matrix work
                    asm volatile("" : "+v"(part[i % RING]) : "v"(bf[nt & 1]), "v"(af[mt]));
#else
                    if constexpr (MATH == 2)
                        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[nt & 1], af[mt], acc[mt][nt], 0, 0, 0, sbe, 0, sae[mt]);
                    else
                    part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                        bf[nt & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
#ifndef DGA_ABL_NODMA
                    if (!LC && i < ISSUE_STEPS) {
#pragma unroll
                        for (int idx = (i * Cfg::LOADS_PER_STAGE) / ISSUE_STEPS;
                             idx < ((i + 1) * Cfg::LOADS_PER_STAGE) / ISSUE_STEPS; ++idx)
                            issue_one(idx, fill, kb + STG - 1);
                    }
#endif
                    // next n-tile's fragment: issued right AFTER this n-tile's first MFMA, so that the (whole-counter)
                    // lgkmcnt wait hipcc places in front of that MFMA never covers reads that were only just issued
#ifndef DGA_ABL_NOLDS
                    if (mt == 0 && nt + 1 < TN) {
                        const int boff = ((nt + 1) >> 1) * 4096 + ((nt + 1) & 1) * 512;
                        const v4i lo = *(const v4i *)(st + b_off0 + boff);
                        const v4i hi = *(const v4i *)(st + b_off1 + boff);
                        bf[(nt + 1) & 1] = v8i{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                    }
#else
                    if (mt == 0 && nt + 1 < TN) { bf[(nt + 1) & 1] = bf[nt & 1]; asm volatile("" : "+v"(bf[(nt + 1) & 1])); }
#endif
                }
                if (MATH != 2 && i == LAG) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) s[mt] *= sfb_v;  // two-level scale: sfa[m,kb] * sfb[n/128,kb]
                }
                if (MATH != 2 && i >= LAG) {
                    const int j = i - LAG, nt = j / TM, mt = j % TM;
                    const v4f pr = part[j % RING];
                    // scalar FMAs on purpose: v_pk_fma_f32 beside MFMAs is slower than two v_fma_f32
                    // (MI355X_MICROARCH "price of one filler beside MFMAs")
#ifdef DGA_ABL_NOFMA
                    asm volatile("" :: "v"(pr), "v"(s[mt]));
#else
                    acc[mt][nt].x = __builtin_fmaf(pr.x, s[mt], acc[mt][nt].x);
                    acc[mt][nt].y = __builtin_fmaf(pr.y, s[mt], acc[mt][nt].y);
                    acc[mt][nt].z = __builtin_fmaf(pr.z, s[mt], acc[mt][nt].z);
                    acc[mt][nt].w = __builtin_fmaf(pr.w, s[mt], acc[mt][nt].w);
#endif
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            DGA_STAMP(4);                            // segment 4: the MFMA / promotion pipeline (+ refill DMA issue)
            stage = stage + 1 == STG ? 0 : stage + 1;
            fill = fill + 1 == STG ? 0 : fill + 1;
        }
        DGA_STAMP_CLOCK(6, 7);
        DGA_STAMP_FLUSH();
        loop_clock.tick();
        loop_clock.flush(p.stamps, blockIdx.x * (NT / 64) + wave, lane);
        epilogue(acc);
    }
}

}  // namespace dga
