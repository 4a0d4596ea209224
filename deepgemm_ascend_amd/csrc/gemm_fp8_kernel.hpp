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
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dga {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

struct GemmParams {
    const uint8_t *a;      // [G][m_rows][lda] e4m3fn bytes
    const float *sfa;      // [G][m_rows][kb_n]
    const uint8_t *b;      // [G][n][ldb]
    const float *sfb;      // [G][nb_n][kb_n]
    uint16_t *out;         // [G][m_rows][ldc] bf16 bits
    const int32_t *masked_m;  // grouped: device int32[G]; dense: nullptr
    int m;                 // dense: M; grouped: m_max (rows allocated per group)
    int n, k, kb_n, nb_n;
    int64_t lda, ldb, ldc;       // row strides in elements
    int64_t a_gs, b_gs, c_gs;    // group strides in elements
    int64_t sfa_gs, sfb_gs;
    int tiles_m, tiles_n, groups;
    int raster_group;            // tile-rows walked together (swizzleOffset analogue, tiling_params.h:63)
    int xcd_remap;               // 1: contiguous tile chunk per XCD (blocks b, b+8, ... share an XCD)
    unsigned long long *stamps;  // diagnostic builds only (-DDGA_STAMPS): per-wave segment cycle sums
};

template <int BM, int BN, int WM, int WN>
struct GemmCfg {
    static constexpr int kBM = BM, kBN = BN, kWM = WM, kWN = WN;
    static constexpr int NT = WM * WN * 64;
    static constexpr int TM = BM / WM / 16;  // m-tiles (16 rows) per wave
    static constexpr int TN = BN / WN / 16;  // n-tiles per wave (even)
    // Waves that issue the LDS-DMA.  (Giving all of it to the first-dispatched half of an 8-wave workgroup -- the
    // half that wins every MFMA arbitration -- evens the two halves out but measured 2-4 % slower overall, r01.)
    static constexpr int DMA_WAVES = WM * WN;
    static constexpr int DNT = DMA_WAVES * 64;  // threads that issue DMA
    static constexpr int A_ROWS = BM * 8 >= DNT ? BM : DNT / 8;  // tiny BM: pad the image to whole wave-instructions
    static constexpr int A_BYTES = A_ROWS * 128;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int SC_SLOTS = ((BM + 8 + DNT - 1) / DNT) * DNT;  // sfa rows, then sfb entries, padded
    static constexpr int SC_BYTES = SC_SLOTS * 4;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES + SC_BYTES;
    static constexpr int STAGES = 2;
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_ITERS = A_ROWS * 8 / DNT;
    static constexpr int B_ITERS = BN * 8 / DNT;
    static constexpr int SC_ITERS = SC_SLOTS / DNT;
    static constexpr int LOADS_PER_STAGE = A_ITERS + B_ITERS + SC_ITERS;
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 32) == 0, "wave tile");
    static_assert(BN % 128 == 0 && BN / WN <= 128 && 128 % (BN / WN) == 0, "a wave's n range lies in one 128-wide scale block");
    static_assert((A_ROWS * 8) % DNT == 0 && (BN * 8) % DNT == 0, "whole wave-instructions per tile");
    static_assert(BN / 128 + (BN % 128 != 0) <= 8, "sfb slots");
};

// LDS image: row r of a tile is 128 bytes = 8 chunks of 16 B; chunk c is stored at
// chunk position c ^ x(r).  x is chosen per operand so that the 16-lane groups of
// ds_read_b128 touch 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz_a(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_b(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Buffer descriptor (raw, stride 0): base address, byte extent, DATA_FORMAT=32 flags word as in the guide's T8.
__device__ __forceinline__ v4i make_rsrc(const void *base, int64_t bytes)
{
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const uint32_t n = bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)n);
    r.w = 0x00020000;
    return r;
}

// LDS-DMA, 16 B per lane: LDS[m0 + 16*lane] = buffer[voff + soff .. +16).  M0 is written in the same statement
// that uses it; `s_nop 4` covers a descriptor word that was produced by v_readfirstlane just before.
// `on` (wave-uniform) skips the instruction with a scalar branch INSIDE the statement, so the compiler's
// scheduling region around the MFMA pipeline stays one straight line.
__device__ __forceinline__ void dma16(uint32_t voff, v4i rsrc, uint32_t soff, uint32_t lds_addr, int on)
{
    asm volatile("s_cmp_eq_u32 %4, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %3\n\ts_nop 4\n\t"
                 "buffer_load_dwordx4 %0, %1, %2 offen lds\n1:"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr), "s"(on) : "memory", "scc");
}
// LDS-DMA, 4 B per lane from a per-lane 64-bit address (the strided scale gather)
__device__ __forceinline__ void dma4(const void *src, uint32_t lds_addr, int on)
{
    asm volatile("s_cmp_eq_u32 %2, 0\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                 "global_load_lds_dword %0, off\n1:"
                 :: "v"(src), "s"(lds_addr), "s"(on) : "memory", "scc");
}

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
#define DGA_STAMP_FLUSH()                                                                       \
    do {                                                                                        \
        if (p.stamps && lane == 0)                                                              \
            for (int q = 0; q < 8; ++q) p.stamps[((size_t)blockIdx.x * (NT / 64) + wave) * 8 + q] = st_seg[q]; \
    } while (0)
#else
#define DGA_STAMP_DECL
#define DGA_STAMP_CLOCK(a, b) do { } while (0)
#define DGA_STAMP_START() do { } while (0)
#define DGA_STAMP(i) do { } while (0)
#define DGA_STAMP_FLUSH() do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class Cfg>
__global__ void __launch_bounds__(Cfg::NT) gemm_fp8_blockscaled_nt_kernel(const GemmParams p)
{
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN;
    constexpr int NT = Cfg::NT, TM = Cfg::TM, TN = Cfg::TN;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // ---- tile id: XCD-aware remap (blocks b, b+8, ... share an XCD and its L2), then
    //      a grouped raster so that an XCD's consecutive tiles share A and B panels.
    const int nwg = gridDim.x;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = p.xcd_remap ? (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3) : bid;
    }
    const int tiles_per_group = p.tiles_m * p.tiles_n;
    const int g = tile / tiles_per_group;
    int t_in = tile - g * tiles_per_group;
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
    if (m0 >= M) return;  // empty expert / fully masked tile: nothing read, nothing written

    const uint8_t *A = p.a + (int64_t)g * p.a_gs;
    const uint8_t *B = p.b + (int64_t)g * p.b_gs;
    const float *SFA = p.sfa + (int64_t)g * p.sfa_gs;
    const float *SFB = p.sfb + (int64_t)g * p.sfb_gs;
    uint16_t *C = p.out + (int64_t)g * p.c_gs;

    // ---- per-thread LDS-DMA sources.  Chunk id c = it*NT + tid lands at LDS byte 16*c (wave-uniform base +
    //      16*lane); it holds source chunk (c&7) ^ x(row) of row c>>3.  A and B are read through buffer
    //      descriptors based at this tile's first row: the per-lane part is a 32-bit byte offset (row*ld + col), the
    //      k advance rides in the scalar offset, and a lane whose chunk lies beyond K is sent out of range, for
    //      which the hardware stores zeros (no branch, no zero page).
    //      DNT is a multiple of 64*4, so (c&7) and x(row) -- hence col -- do not depend on `it`.
    constexpr int DNT = Cfg::DNT;
    const int dma_on = __builtin_amdgcn_readfirstlane(wave < Cfg::DMA_WAVES ? 1 : 0);
    const int dtid = tid & (DNT - 1);
    constexpr uint32_t kOutOfRange = 0x80000000u;  // > num_records (host guarantees tile extents < 2^31)
    const int a_col = ((dtid & 7) ^ swz_a(dtid >> 3)) * 16;
    const int b_col = ((dtid & 7) ^ swz_b(dtid >> 3)) * 16;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::A_ITERS; ++it) {
        const int row = (it * DNT + dtid) >> 3;
        a_voff[it] = (uint32_t)min(row, M - 1 - m0) * (uint32_t)p.lda + a_col;
    }
#pragma unroll
    for (int it = 0; it < Cfg::B_ITERS; ++it) {
        const int row = (it * DNT + dtid) >> 3;
        b_voff[it] = (uint32_t)min(row, p.n - 1 - n0) * (uint32_t)p.ldb + b_col;
    }
    const v4i a_rsrc = make_rsrc(A + (int64_t)m0 * p.lda, (int64_t)(M - m0) * p.lda);
    const v4i b_rsrc = make_rsrc(B + (int64_t)n0 * p.ldb, (int64_t)(p.n - n0) * p.ldb);
    // scale slots: [0,BM) = sfa rows of this tile, [BM, BM+8) = sfb blocks of this tile, rest = padding
    const float *sc_src[Cfg::SC_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::SC_ITERS; ++it) {
        const int s = it * DNT + dtid;
        if (s < BM) {
            sc_src[it] = SFA + (int64_t)min(m0 + s, M - 1) * p.kb_n;
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
    auto issue_one = [&](int idx, int stage, int kb) {
        const uint32_t sa = lds0 + stage * Cfg::STAGE_BYTES;
        const uint32_t sb = sa + Cfg::A_BYTES;
        const uint32_t ss = sb + Cfg::B_BYTES;
        const int k0 = kb * 128;
        if (idx < Cfg::A_ITERS) {
            const int it = idx;
            const uint32_t voff = (k0 + a_col < p.k) ? a_voff[it] : kOutOfRange;
            dma16(voff, a_rsrc, (uint32_t)k0, sa + (it * DNT + wave * 64) * 16, dma_on);
        } else if (idx < Cfg::A_ITERS + Cfg::B_ITERS) {
            const int it = idx - Cfg::A_ITERS;
            const uint32_t voff = (k0 + b_col < p.k) ? b_voff[it] : kOutOfRange;
            dma16(voff, b_rsrc, (uint32_t)k0, sb + (it * DNT + wave * 64) * 16, dma_on);
        } else {
            const int it = idx - Cfg::A_ITERS - Cfg::B_ITERS;
            dma4(sc_src[it] + min(kb, p.kb_n - 1), ss + (it * DNT + wave * 64) * 4, dma_on);
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

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    // ---- main loop.  ONE barrier per k block: passing it means (a) every wave's DMA of this stage has landed
    //      (each waited vmcnt(0) first) and (b) every wave has left the previous k block, i.e. the other stage is
    //      free -- so its refill is issued from inside this k block's MFMA pipeline, one DMA wave-instruction every
    //      few MFMAs.  (Issued in a burst at the top, the 9 DMA instructions cost a wave 800-1600 cycles of blocked
    //      issue per k block: the vector-memory path takes 64 B/clk/CU; measured with the -DDGA_STAMPS build.)
    const int KB = p.kb_n;
    DGA_STAMP_DECL
#pragma unroll
    for (int idx = 0; idx < Cfg::LOADS_PER_STAGE; ++idx) issue_one(idx, 0, 0);
    DGA_STAMP_START();
    DGA_STAMP_CLOCK(6, 7);   // slots 6/7: shader-clock and 100 MHz real-time ticks across the main loop
    for (int kb = 0; kb < KB; ++kb) {
        const int stage = kb & 1;
        wait_vmcnt<0>();
        DGA_STAMP(1);                            // segment 1: vmcnt wait
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");           // no LDS read may be hoisted above the barrier
        DGA_STAMP(2);                            // segment 2: barrier "stage ready, other stage free"

        const uint8_t *st = smem + stage * Cfg::STAGE_BYTES;
        // Fragment reads are ordered so that the first MFMA waits only for ITS operands (B n-tile 0, A m-tile 0):
        // all 8 waves hit the LDS at once here, and a wave that waited for its whole 15-read burst would idle the
        // matrix pipe for ~450 cycles per k block.  The scale reads come last; they are first needed LAG steps later.
        constexpr int STEPS = TM * TN, LAG = 3, RING = LAG + 1;
        constexpr int ISSUE_STEPS = (STEPS * 5) / 8 > 0 ? (STEPS * 5) / 8 : 1;  // refill DMA rides on the first 5/8 of the steps
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
        DGA_STAMP(3);                            // segment 3: first fragments + scales out of LDS
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < STEPS + LAG; ++i) {
            if (i < STEPS) {
                const int nt = i / TM, mt = i % TM;
                part[i % RING] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(
                    bf[nt & 1], af[mt], v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (i < ISSUE_STEPS) {
#pragma unroll
                    for (int idx = (i * Cfg::LOADS_PER_STAGE) / ISSUE_STEPS;
                         idx < ((i + 1) * Cfg::LOADS_PER_STAGE) / ISSUE_STEPS; ++idx)
                        issue_one(idx, stage ^ 1, kb + 1);
                }
                // next n-tile's fragment: issued right AFTER this n-tile's first MFMA, so that the (whole-counter)
                // lgkmcnt wait hipcc places in front of that MFMA never covers reads that were only just issued
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
                // scalar FMAs on purpose: v_pk_fma_f32 beside MFMAs is slower than two v_fma_f32
                // (MI355X_MICROARCH "price of one filler beside MFMAs")
                acc[mt][nt].x = __builtin_fmaf(pr.x, s[mt], acc[mt][nt].x);
                acc[mt][nt].y = __builtin_fmaf(pr.y, s[mt], acc[mt][nt].y);
                acc[mt][nt].z = __builtin_fmaf(pr.z, s[mt], acc[mt][nt].z);
                acc[mt][nt].w = __builtin_fmaf(pr.w, s[mt], acc[mt][nt].w);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        DGA_STAMP(4);                            // segment 4: the MFMA / promotion pipeline (+ refill DMA issue)
    }
    DGA_STAMP_CLOCK(6, 7);
    DGA_STAMP_FLUSH();

    // ---- epilogue: lane owns row m, columns n0w + 32*j + 8*(lane>>4) + [0,8)
    const int m_row = m0 + wm * (BM / Cfg::kWM) + li;
    const int n_base = n0 + wn * (BN / WN) + 8 * kg;
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

// Generic kernel: any K (also K % 16 != 0), any strides.  One thread per output
// element, fp32 running sums in the oracle's order.  Used only where the LDS-DMA
// kernel's 16-byte chunking does not apply.
__device__ __forceinline__ float e4m3fn_to_f32(uint8_t v)
{
    const uint32_t e = (v >> 3) & 15u, mnt = v & 7u;
    float r;
    if (e == 0) r = (float)mnt * 0.001953125f;  // subnormal: mnt/8 * 2^-6
    else if (e == 15u && mnt == 7u) r = __builtin_nanf("");
    else r = __uint_as_float(((e + 120u) << 23) | (mnt << 20));
    return (v & 0x80) ? -r : r;
}

__global__ void __launch_bounds__(256) gemm_fp8_blockscaled_nt_generic_kernel(const GemmParams p)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = e4m3fn_to_f32((uint8_t)threadIdx.x);
    __syncthreads();
    const int g = blockIdx.z;
    const int M = p.masked_m ? min(p.masked_m[g], p.m) : p.m;
    const int n = blockIdx.x * 16 + (threadIdx.x & 15);
    const int m = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (m >= M || n >= p.n) return;
    const uint8_t *ar = p.a + (int64_t)g * p.a_gs + (int64_t)m * p.lda;
    const uint8_t *br = p.b + (int64_t)g * p.b_gs + (int64_t)n * p.ldb;
    const float *sa = p.sfa + (int64_t)g * p.sfa_gs + (int64_t)m * p.kb_n;
    const float *sb = p.sfb + (int64_t)g * p.sfb_gs + (int64_t)(n / 128) * p.kb_n;
    float acc = 0.f;
    for (int kb = 0; kb < p.kb_n; ++kb) {
        const int k0 = kb * 128, k1 = min(p.k, k0 + 128);
        float part = 0.f;
        for (int k = k0; k < k1; ++k) part += lut[ar[k]] * lut[br[k]];
        acc += part * (sa[kb] * sb[kb]);
    }
    const v2bf h = __builtin_convertvector(v2f{acc, 0.f}, v2bf);
    p.out[(int64_t)g * p.c_gs + (int64_t)m * p.ldc + n] = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
}

}  // namespace dga
