// 16-bit (bf16 / fp16) GEMM with fp32 output for the framework's own entry points -- run_mmad_rtc / run_mmad_bench
// (/root/reference/deep_gemm_ascend/framework/csrc/jit_kernels/impls/gemm.hpp:68-111, gemm_bench.hpp:49-113; device
// algorithm /root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:123-369: fp32 accumulate, fp32 out).
//
// The reference kernel takes y as [K,N] and re-lays both operands on the way into L1 (ND->NZ DataCopy, :231-260).
// Here the re-layout of y is a separate transposing pass into the caller's workspace (yT [N,Kp], Kp = K rounded up
// to 64, zero filled), after which both operands are K-contiguous and the GEMM is the same LDS-DMA / swizzled-image /
// one-barrier-per-k-step structure as the fp8 kernel (gemm_fp8_kernel.hpp), with
//   k step = 64 elements = 128 bytes per row  (identical LDS image, swizzle and DMA addressing),
//   two v_mfma_f32_16x16x32_{bf16,f16} per 16x16 tile per k step accumulating in place (no promotion FMAs),
//   fp32 epilogue: a lane owns 4 consecutive n of one row -> 16-byte stores.
#pragma once
#include "dga_device_common.hpp"

namespace dga {

typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef _Float16 v8f16 __attribute__((ext_vector_type(8)));

struct B16Params {
    const uint16_t *x;   // [batch][m][ldx]   (K-contiguous rows)
    const uint16_t *yt;  // [batch][n][ldy]   (transposed y: K-contiguous rows)
    float *z;            // [batch][m][n]
    int m, n, k;         // k = padded K (multiple of 64)
    int64_t ldx, ldy;
    int64_t x_bs, y_bs, z_bs;
    int tiles_m, tiles_n, raster_group;
    int batch;           // z / x / yt matrices
    int splitk;          // > 1: the grid is splitk x batch x tiles (split-major); split s covers k steps [s*ks_per_split, +)
    int ks_per_split;    //      and writes its fp32 partial tile to slab s of `partial` ([splitk][batch][m][n])
    float *partial;
    uint16_t *z16;       // OUT16 builds: the output in the inputs' 16-bit type ([batch][m][n]); z is unused then
    int launch_tiles;    // > 0: the grid holds this many tiles (batch 1, no split-K), not the whole raster
    int tail_begin, tail_sub;   // tail_sub = sm | sn << 8 != 0: this launch's tiles are SUB-tiles (sm x sn per parent) of the parent
                                // raster's tiles [tail_begin, ...); tiles_m / tiles_n / raster_group then describe the PARENT raster
};

template <bool BF16>
__device__ __forceinline__ v4f mfma_b16(v4i a, v4i b, v4f c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf16, a), __builtin_bit_cast(v8bf16, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8f16, a), __builtin_bit_cast(v8f16, b), c, 0, 0, 0);
}

// PP = 0: one barrier per k step in front of the fragment reads (every tile).
// PP = 2: continuous pipeline (256x256, 8 waves), the 16-bit form of the fp8 kernel's schedule: the k-step boundary
//         disappears from the MFMA stream (fragments of the next step are read under the last MFMAs of this one).
// NN = false: the y operand comes transposed (yT [N][Kp], K-contiguous rows, written by transpose_b16_kernel).
// NN = true:  y is read where it lies, [K][N] with N contiguous -- no pre-pass, no workspace.  The LDS image of the
//             tile is then [64 k rows][BN columns] and the MFMA operand (8 consecutive k of one n per lane) is produced
//             by the hardware transposing read ds_read_b64_tr_b16: per 16-lane group, lane 4q+p supplies the address of
//             row q / columns 4p..4p+3 of a 4 x 16 block and lane i receives column i.  Group g takes rows 8g+q (+4 for
//             the second read), so an operand is two reads.  Rows are BN*2 bytes (a multiple of the 256-byte bank row),
//             so the 16-byte chunk c of row k is stored at chunk c ^ 2*h(k), h(k) = (k&3) | ((k>>3)&1)<<2: the 8 rows
//             a 32-lane half touches then fall into 8 different 32-byte groups of one bank row (conflict-free).  The
//             n index of MFMA row i is simply 16*nt + i here (the fp32 stores need no pairing of n-tiles).
//             Requires K % 64 == 0, N % 8 == 0 and 16-byte aligned bases (the host falls back to NN = false otherwise).
// OUT16 = true: the result is rounded (RNE) to the inputs' 16-bit type and stored with 16-byte rows segments (two
//             n-tiles = 8 consecutive n per lane, as in the fp8 kernel) -- the dtype contract of the aclnn operator
//             (catlass_dynamic_matmul.cpp:37-46: out dtype = input dtype).  NT (NN = false) only.
template <class Cfg, bool BF16, int PP = 0, bool NN = false, bool OUT16 = false>
__global__ void __launch_bounds__(Cfg::NT) gemm_b16_nt_f32_kernel(const B16Params p)
{
    static_assert(!(OUT16 && NN), "16-bit output is built for the NT operand layout");
    constexpr int BM = Cfg::kBM, BN = Cfg::kBN, WN = Cfg::kWN;
    constexpr int NT = Cfg::NT, TM = Cfg::TM, TN = Cfg::TN, DNT = Cfg::DNT;
    constexpr int STAGE = Cfg::A_BYTES + Cfg::B_BYTES;  // no scale slots
    constexpr int NL = Cfg::A_ITERS + Cfg::B_ITERS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x;
    int tile;
    {
        const int bid = blockIdx.x, xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
        tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int per_batch = p.tiles_m * p.tiles_n;
    const int split = p.splitk > 1 ? tile / (per_batch * p.batch) : 0;
    tile -= split * per_batch * p.batch;
    const int bi = p.tail_sub ? 0 : tile / per_batch;
    const int sm = p.tail_sub & 255, sn = p.tail_sub >> 8, nsub = max(1, sm * sn);
    const int sub = p.tail_sub ? tile % nsub : 0;
    const int t_in = p.tail_sub ? p.tail_begin + tile / nsub : tile - bi * per_batch;
    int tm, tn;
    {
        const int gm = p.raster_group, per = gm * p.tiles_n, band = t_in / per, first = band * gm;
        const int rows = min(p.tiles_m - first, gm), loc = t_in - band * per;
        tm = first + loc % rows;
        tn = loc / rows;
    }
    const int m0 = p.tail_sub ? (sm * tm + sub % sm) * BM : tm * BM, n0 = p.tail_sub ? (sn * tn + sub / sm) * BN : tn * BN;
    if (p.tail_sub && (m0 >= p.m || n0 >= p.n)) return;   // a quarter tile beyond the matrix edge
    const uint8_t *X = (const uint8_t *)(p.x + (int64_t)bi * p.x_bs);
    const uint8_t *Y = (const uint8_t *)(p.yt + (int64_t)bi * p.y_bs);
    float *Z = (p.splitk > 1 ? p.partial + (int64_t)split * p.batch * p.z_bs : p.z) + (int64_t)bi * p.z_bs;
    const int64_t ldxb = p.ldx * 2, ldyb = p.ldy * 2;  // row strides in bytes

    const int dtid = tid & (DNT - 1);
    const int a_col = ((dtid & 7) ^ swz_a(dtid >> 3)) * 16, b_col = ((dtid & 7) ^ swz_b(dtid >> 3)) * 16;
    uint32_t a_voff[Cfg::A_ITERS], b_voff[Cfg::B_ITERS];
#pragma unroll
    for (int it = 0; it < Cfg::A_ITERS; ++it)
        a_voff[it] = (uint32_t)min((it * DNT + dtid) >> 3, p.m - 1 - m0) * (uint32_t)ldxb + a_col;
    constexpr int CPR = BN / 8;  // NN: 16-byte chunks per LDS row of the y tile
#pragma unroll
    for (int it = 0; it < Cfg::B_ITERS; ++it) {
        if constexpr (NN) {
            const int cid = it * DNT + dtid, kr = cid / CPR, cs = cid % CPR;
            const int c = cs ^ (2 * ((kr & 3) | (((kr >> 3) & 1) << 2)));  // swizzle on the source side
            b_voff[it] = (uint32_t)kr * (uint32_t)ldyb + (uint32_t)c * 16u;
        } else {
            b_voff[it] = (uint32_t)min((it * DNT + dtid) >> 3, p.n - 1 - n0) * (uint32_t)ldyb + b_col;
        }
    }
    const v4i a_rsrc = make_rsrc(X + (int64_t)m0 * ldxb, (int64_t)(p.m - m0) * ldxb);
    const v4i b_rsrc = NN ? make_rsrc(Y + (int64_t)n0 * 2, (int64_t)p.k * ldyb - (int64_t)n0 * 2)
                          : make_rsrc(Y + (int64_t)n0 * ldyb, (int64_t)(p.n - n0) * ldyb);
    const uint32_t b_kstep = NN ? (uint32_t)(64 * ldyb) : 128u;  // bytes one k step advances in the y operand
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    auto issue_one = [&](int idx, int stage, int ks) {
        const uint32_t sa = lds0 + stage * STAGE + wave * 1024;
        if (idx < Cfg::A_ITERS) dma16(a_voff[idx], a_rsrc, (uint32_t)(ks * 128), sa + idx * DNT * 16);
        else dma16(b_voff[idx - Cfg::A_ITERS], b_rsrc, (uint32_t)ks * b_kstep, sa + Cfg::A_BYTES + (idx - Cfg::A_ITERS) * DNT * 16);
    };

    const int li = lane & 15, kg = lane >> 4;
    const int a_row = wm * (BM / Cfg::kWM) + li;
    const int a_off0 = a_row * 128 + ((kg ^ swz_a(a_row)) * 16), a_off1 = a_row * 128 + (((kg + 4) ^ swz_a(a_row)) * 16);
    const int b_row = wn * (BN / WN) + 8 * (li >> 2) + (li & 3);
    const int b_off0 = Cfg::A_BYTES + b_row * 128 + ((kg ^ swz_b(b_row)) * 16);
    const int b_off1 = Cfg::A_BYTES + b_row * 128 + (((kg + 4) ^ swz_b(b_row)) * 16);
    // NN: lane = 16g + 4q + p supplies row 8g + q, columns (wave's n base) + 16 nt + 4p; chunk swizzle h = q | (g&1)<<2
    const int tq = (lane >> 2) & 3, tp = lane & 3;
    const int nn_h = tq | ((kg & 1) << 2);
    const int nn_base = Cfg::A_BYTES + (8 * kg + tq) * (BN * 2) + (tp >> 1) * 16 + 8 * (tp & 1);
    const int nn_cb = wn * (BN / WN) / 16;  // the wave's first 32-byte chunk pair (n-tile nt is pair nn_cb + nt)
    // operand of n-tile nt, k half h (k = 32h .. 32h+31 of the step) out of stage `st`
    auto load_b = [&](const uint8_t *st, int nt, int h) -> v4i {
        if constexpr (NN) {
            typedef short v4s __attribute__((ext_vector_type(4)));
            typedef __attribute__((address_space(3))) v4s *lds_v4s;
            const uint8_t *a0 = st + nn_base + (((nn_cb + nt) ^ nn_h) << 5) + h * 32 * (BN * 2);
            const v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a0));
            const v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(a0 + 4 * (BN * 2)));
            const v2i l2 = __builtin_bit_cast(v2i, lo), h2 = __builtin_bit_cast(v2i, hi);
            return v4i{l2.x, l2.y, h2.x, h2.y};
        } else {
            const int boff = (nt >> 1) * 4096 + (nt & 1) * 512;
            return *(const v4i *)(st + (h ? b_off1 : b_off0) + boff);
        }
    };

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int KS = p.k / 64;
    const int ks_begin = p.splitk > 1 ? split * p.ks_per_split : 0;   // this workgroup's k steps (split-K: one slice)
    const int ks_end = p.splitk > 1 ? min(KS, ks_begin + p.ks_per_split) : KS;
    if constexpr (PP == 2) {
        // ---- continuous pipeline.  One k step = STEPS single MFMAs, ordered (n-tile, k half, m-tile) so that the two
        // MFMAs that chain through one accumulator are TM steps apart.  Stage s = parity of the step inside the slice.
        //   * B fragments rotate through two register sets (next n-tile read at the n-tile's first step); the rotation
        //     continues into the next k step;
        //   * step SB = first step of the last n-tile: vmcnt(0) + the ONE barrier.  Passing it means stage s^1 has landed
        //     everywhere and nobody reads stage s any more (its last read, the last n-tile's B fragment, was issued an
        //     n-tile earlier);
        //   * during the last n-tile every A fragment half is reloaded IN PLACE from stage s^1 right behind its last
        //     MFMA, and the next step's first B fragment is read: the next k step finds its operands in registers;
        //   * refill of stage s (k step ks + 2): TAIL_DMA wave-instructions on the steps behind the barrier, the rest on
        //     the first HEAD_STEPS steps of the next k step -- every batch has more than half a k step to land.
        constexpr int STEPS = TN * 2 * TM, SB = (TN - 1) * 2 * TM;
        constexpr int TAIL_DMA = NL / 2, HEAD_STEPS = (STEPS * 9) / 32;
        static_assert(BM == 256 && BN == 256 && Cfg::kWM == 4 && WN == 2, "continuous schedule: 256x256, 8 waves");
        auto barrier = [&]() {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        };
        v4i af[TM][2], bf[2][2];
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_one(idx, 0, ks_begin);
#pragma unroll
        for (int idx = 0; idx < TAIL_DMA; ++idx) issue_one(idx, 1, ks_begin + 1);
        wait_vmcnt<TAIL_DMA>();
        barrier();
        bf[0][0] = load_b(smem, 0, 0);
        bf[0][1] = load_b(smem, 0, 1);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            af[mt][0] = *(const v4i *)(smem + a_off0 + mt * 2048);
            af[mt][1] = *(const v4i *)(smem + a_off1 + mt * 2048);
        }
        for (int ks = ks_begin; ks < ks_end; ++ks) {
            const int par = (ks - ks_begin) & 1;   // stage of this k step
            const uint8_t *st = smem + par * STAGE;
            const uint8_t *sn = smem + (par ^ 1) * STAGE;
#pragma unroll
            for (int i = 0; i < STEPS; ++i) {
                const int nt = i / (2 * TM), h = (i / TM) & 1, mt = i % TM;
                if (i == SB) {
                    wait_vmcnt<0>();
                    barrier();
                }
                acc[mt][nt] = mfma_b16<BF16>(bf[nt & 1][h], af[mt][h], acc[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
                if (i < HEAD_STEPS) {   // head part of k step ks + 1's refill (stage s^1 is free since the previous barrier)
#pragma unroll
                    for (int j = (i * (NL - TAIL_DMA)) / HEAD_STEPS; j < ((i + 1) * (NL - TAIL_DMA)) / HEAD_STEPS; ++j)
                        issue_one(TAIL_DMA + j, par ^ 1, ks + 1);
                }
                if (i > SB && i <= SB + TAIL_DMA) issue_one(i - SB - 1, par, ks + 2);  // tail part of ks + 2 into this stage
                if (h == 0 && mt == 0) {  // next n-tile's B fragment (wraps into the next k step)
                    const uint8_t *src = nt + 1 < TN ? st : sn;
                    const int nn = nt + 1 < TN ? nt + 1 : 0;
                    bf[(nt + 1) & 1][0] = load_b(src, nn, 0);
                    bf[(nt + 1) & 1][1] = load_b(src, nn, 1);
                }
                if (nt == TN - 1) af[mt][h] = *(const v4i *)(sn + (h ? a_off1 : a_off0) + mt * 2048);  // in-place reload
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        wait_vmcnt<0>();
    } else {
    // STG LDS stages (Cfg::STAGES; 3 for the 128x256 8-wave build: two refills in flight)
    constexpr int STG = Cfg::STAGES;
#pragma unroll
    for (int d = 0; d < STG - 1; ++d)
#pragma unroll
        for (int idx = 0; idx < NL; ++idx) issue_one(idx, d, ks_begin + d);
    int stage = 0, fill = STG - 1;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        wait_vmcnt<(STG - 2) * NL>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const uint8_t *st = smem + stage * STAGE;
        v4i af[TM][2], bf[2][2];
        bf[0][0] = load_b(st, 0, 0);
        bf[0][1] = load_b(st, 0, 1);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            af[mt][0] = *(const v4i *)(st + a_off0 + mt * 2048);
            af[mt][1] = *(const v4i *)(st + a_off1 + mt * 2048);
        }
        // per n-tile: the first-half MFMAs of every m-tile, then the second halves (no back-to-back dependent pair);
        // the refill DMA of the other stage rides on the first 5/8 of the n-tiles
        constexpr int ISSUE_NT = (TN * 5) / 8 > 0 ? (TN * 5) / 8 : 1;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) acc[mt][nt] = mfma_b16<BF16>(bf[nt & 1][0], af[mt][0], acc[mt][nt]);
            if (nt < ISSUE_NT) {
#pragma unroll
                for (int idx = (nt * NL) / ISSUE_NT; idx < ((nt + 1) * NL) / ISSUE_NT; ++idx)
                    issue_one(idx, fill, ks + STG - 1);  // past the last k step: reads the tile's following bytes, unused
            }
            if (nt + 1 < TN) {
                bf[(nt + 1) & 1][0] = load_b(st, nt + 1, 0);
                bf[(nt + 1) & 1][1] = load_b(st, nt + 1, 1);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) acc[mt][nt] = mfma_b16<BF16>(bf[nt & 1][1], af[mt][1], acc[mt][nt]);
        }
        stage = stage + 1 == STG ? 0 : stage + 1;
        fill = fill + 1 == STG ? 0 : fill + 1;
    }
    wait_vmcnt<0>();
    }

    // epilogue: lane owns row m, 4 consecutive n per 16x16 tile (same n permutation as the fp8 kernel)
    const int m_row = m0 + wm * (BM / Cfg::kWM) + li;
    const int n_base = n0 + wn * (BN / WN) + (NN ? 4 : 8) * kg;
    if (OUT16 && p.splitk <= 1) {
        uint16_t *Z16 = p.z16 + (int64_t)bi * p.z_bs;
        const bool v16_ok = ((p.n & 7) == 0) && ((((uintptr_t)Z16) & 15) == 0);
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m_row + mt * 16;
            if (m >= p.m) continue;
            uint16_t *zr = Z16 + (int64_t)m * p.n;
#pragma unroll
            for (int j = 0; j < TN / 2; ++j) {
                const int n = n_base + 32 * j;
                const v4f lo = acc[mt][2 * j], hi = acc[mt][2 * j + 1];
                v4i pk;
                if constexpr (BF16) {
                    pk = v4i{__builtin_bit_cast(int, __builtin_convertvector(v2f{lo.x, lo.y}, v2bf)),
                             __builtin_bit_cast(int, __builtin_convertvector(v2f{lo.z, lo.w}, v2bf)),
                             __builtin_bit_cast(int, __builtin_convertvector(v2f{hi.x, hi.y}, v2bf)),
                             __builtin_bit_cast(int, __builtin_convertvector(v2f{hi.z, hi.w}, v2bf))};
                } else {
                    typedef _Float16 v2h __attribute__((ext_vector_type(2)));
                    pk = v4i{__builtin_bit_cast(int, v2h{(_Float16)lo.x, (_Float16)lo.y}),
                             __builtin_bit_cast(int, v2h{(_Float16)lo.z, (_Float16)lo.w}),
                             __builtin_bit_cast(int, v2h{(_Float16)hi.x, (_Float16)hi.y}),
                             __builtin_bit_cast(int, v2h{(_Float16)hi.z, (_Float16)hi.w})};
                }
                if (v16_ok && n + 8 <= p.n) {
                    *(v4i *)(zr + n) = pk;
                } else {
                    const uint16_t *e = (const uint16_t *)&pk;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (n + q < p.n) zr[n + q] = e[q];
                }
            }
        }
        return;
    }
    const bool v_ok = ((p.n & 3) == 0) && ((((uintptr_t)Z) & 15) == 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = m_row + mt * 16;
        if (m >= p.m) continue;
        float *zr = Z + (int64_t)m * p.n;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int n = NN ? n_base + 16 * nt : n_base + 32 * (nt >> 1) + 4 * (nt & 1);
            if (v_ok && n + 4 <= p.n) {
                *(v4f *)(zr + n) = acc[mt][nt];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (n + q < p.n) zr[n + q] = acc[mt][nt][q];
            }
        }
    }
}

#ifndef DGA_B16_TILE_KERNEL_ONLY   // (a second translation unit that includes this header for the tile kernel and its types only: dga_b16_w4.hip)
// y [K][N] (16-bit) -> yT [N][kp], kp = K rounded up to 64, zero filled: 64 x 64 tiles through LDS.  Both global
// sides move 16 bytes per lane on full 128-byte row segments (8 lanes per row); the transposition happens in the LDS
// reads (eight 16-bit reads of one column, conflict-free on the 66-element pitch).  `vec` = N % 8 == 0 and both bases
// 16-byte aligned; otherwise the loads fall back to element accesses.
__global__ void __launch_bounds__(256) transpose_b16_kernel(const uint16_t *y, uint16_t *yt, int k, int n, int kp,
                                                            int64_t y_bs, int64_t yt_bs, int vec)
{
    constexpr int PITCH = 66;
    __shared__ __attribute__((aligned(16))) uint16_t tile[64 * PITCH];
    const uint16_t *src = y + (int64_t)blockIdx.z * y_bs;
    uint16_t *dst = yt + (int64_t)blockIdx.z * yt_bs;
    const int k0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int t = threadIdx.x, r8 = t >> 3, c8 = (t & 7) * 8;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int kk = k0 + r8 + it * 32, nn = n0 + c8;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (kk < k) {
            const uint16_t *row = src + (int64_t)kk * n;
            if (vec && nn + 8 <= n) {
                const v4i q = *(const v4i *)(row + nn);
                w[0] = (uint32_t)q.x; w[1] = (uint32_t)q.y; w[2] = (uint32_t)q.z; w[3] = (uint32_t)q.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (nn + j < n) w[j >> 1] |= (uint32_t)row[nn + j] << (16 * (j & 1));
            }
        }
        uint32_t *lw = (uint32_t *)(tile + (r8 + it * 32) * PITCH + c8);  // 4-byte aligned (PITCH and c8 are even)
        lw[0] = w[0]; lw[1] = w[1]; lw[2] = w[2]; lw[3] = w[3];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int nl = r8 + it * 32, nn = n0 + nl, kk = k0 + c8;
        if (nn >= n || kk >= kp) continue;
        uint32_t w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            w[j] = (uint32_t)tile[(c8 + 2 * j) * PITCH + nl] | ((uint32_t)tile[(c8 + 2 * j + 1) * PITCH + nl] << 16);
        *(v4i *)(dst + (int64_t)nn * kp + kk) = v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};  // kp % 64 == 0, base 256-B aligned
    }
}

// split-K combine of the 16-bit path: z = sum_s slab[s] in fp32, s ascending (deterministic)
__global__ void __launch_bounds__(256) splitk_reduce_f32_kernel(const float *partial, float *z, int64_t total, int splitk)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= total) return;
    if (i + 4 <= total && ((total & 3) == 0) && ((((uintptr_t)z) & 15) == 0)) {
        v4f a = *(const v4f *)(partial + i);
        for (int s = 1; s < splitk; ++s) a += *(const v4f *)(partial + (int64_t)s * total + i);
        *(v4f *)(z + i) = a;
    } else {
        for (int q = 0; q < 4 && i + q < total; ++q) {
            float a = partial[i + q];
            for (int s = 1; s < splitk; ++s) a += partial[(int64_t)s * total + i + q];
            z[i + q] = a;
        }
    }
}

#endif  // DGA_B16_TILE_KERNEL_ONLY
// split-K combine with 16-bit output (the operator's dtype contract): out = round16( sum_s slab[s] ), s ascending
template <bool BF16>
__global__ void __launch_bounds__(256) splitk_reduce_16_kernel(const float *partial, uint16_t *z, int64_t total, int splitk)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    float a = partial[i];
    for (int s = 1; s < splitk; ++s) a += partial[(int64_t)s * total + i];
    if constexpr (BF16) {
        const v2bf h = __builtin_convertvector(v2f{a, 0.f}, v2bf);
        z[i] = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
    } else {
        z[i] = __builtin_bit_cast(uint16_t, (_Float16)a);
    }
}

// element-wise 16-bit NT GEMM (any K, any alignment; fp32 running sum in k order): the route of last resort of the
// operator entry when there is no workspace to pad into
template <bool BF16>
__global__ void __launch_bounds__(256) gemm_b16_nt_generic_kernel(const uint16_t *a, const uint16_t *b, uint16_t *out, int m,
                                                                  int n, int k)
{
    const int col = blockIdx.x * 16 + (threadIdx.x & 15), row = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (row >= m || col >= n) return;
    const uint16_t *ar = a + (int64_t)row * k, *br = b + (int64_t)col * k;
    float acc = 0.f;
    for (int i = 0; i < k; ++i) {
        float x, y;
        if constexpr (BF16) { x = __uint_as_float((uint32_t)ar[i] << 16); y = __uint_as_float((uint32_t)br[i] << 16); }
        else { x = (float)__builtin_bit_cast(_Float16, ar[i]); y = (float)__builtin_bit_cast(_Float16, br[i]); }
        acc = __builtin_fmaf(x, y, acc);
    }
    if constexpr (BF16) {
        const v2bf h = __builtin_convertvector(v2f{acc, 0.f}, v2bf);
        out[(int64_t)row * n + col] = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
    } else {
        out[(int64_t)row * n + col] = __builtin_bit_cast(uint16_t, (_Float16)acc);
    }
}

}  // namespace dga
