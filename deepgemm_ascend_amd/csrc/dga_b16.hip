// The framework's 16-bit path: z (f32) = x . y with x [M,K], y [K,N] row-major bf16 / fp16 ("NN").
// Replaces the device side of run_mmad_rtc / run_mmad_bench
//   /root/reference/deep_gemm_ascend/framework/csrc/jit_kernels/impls/gemm.hpp:68-111   (batched, bf16)
//   /root/reference/deep_gemm_ascend/framework/csrc/jit_kernels/impls/gemm_bench.hpp:49-113 (single, fp16, params[28])
// whose generated AscendC kernel (framework/csrc/jit/generate_code.hpp:123-369) accumulates in fp32 and writes fp32.
//
// Two device paths, both on v_mfma_f32_16x16x32_{bf16,f16}:
//   * with a workspace (dga_run_mmad_*_ws; what the Python entry points use): y is transposed into the workspace
//     and the LDS-DMA tile kernel of gemm_b16_kernel.hpp runs (0.9-1.0 PFLOP/s at 4096^3..8192^3, r01);
//   * without one (the plain C entry points): each lane gathers its B fragment straight from global memory
//     (8 two-byte loads down a column) -- no LDS, no transposition pass; correct, slow.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <cstdint>
#include <mutex>

#include "dga_hip.h"
#include "dga_internal.hpp"
#include "gemm_b16_kernel.hpp"
#include "gemm_b16_wsk_kernel.hpp"

namespace dga {

int launch_b16_w4(const B16Params &p, bool bf16, hipStream_t stream);   // dga_b16_w4.hip

typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef uint16_t v8u __attribute__((ext_vector_type(8)));

struct B16DirectParams {
    const uint16_t *x, *y;
    float *z;
    int m, n, k;
    int64_t x_bs, y_bs, z_bs;  // batch strides (elements)
};

template <bool BF16>
__device__ __forceinline__ v4f mfma16(v8u a, v8u b, v4f c)
{
    if constexpr (BF16)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a), __builtin_bit_cast(v8h, b), c, 0, 0, 0);
}

// workgroup = 4 waves (2 x 2), wave tile 64 x 64 = 4 x 4 MFMA tiles, K step 32
template <bool BF16>
__global__ void __launch_bounds__(256) mmad_nn_f32_kernel(const B16DirectParams p)
{
    constexpr int TM = 4, TN = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, kg = lane >> 4;
    const int m0 = blockIdx.y * 128 + (wave >> 1) * 64, n0 = blockIdx.x * 128 + (wave & 1) * 64;
    const uint16_t *X = p.x + (int64_t)blockIdx.z * p.x_bs;
    const uint16_t *Y = p.y + (int64_t)blockIdx.z * p.y_bs;
    float *Z = p.z + (int64_t)blockIdx.z * p.z_bs;
    if (m0 >= p.m || n0 >= p.n) return;

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const bool x_vec = ((p.k & 7) == 0) && ((((uintptr_t)X) & 15) == 0);
    for (int k0 = 0; k0 < p.k; k0 += 32) {
        const int kk = k0 + 8 * kg;  // this lane's 8 consecutive k
        v8u af[TM], bf[TN];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int r = m0 + mt * 16 + li;
            v8u v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (r < p.m) {
                const uint16_t *src = X + (int64_t)r * p.k + kk;
                if (x_vec && kk + 8 <= p.k) {
                    v = *(const v8u *)src;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if (kk + j < p.k) v[j] = src[j];
                }
            }
            af[mt] = v;
        }
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int c = n0 + nt * 16 + li;
            v8u v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c < p.n) {
                const uint16_t *src = Y + (int64_t)kk * p.n + c;
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (kk + j < p.k) v[j] = src[(int64_t)j * p.n];
            }
            bf[nt] = v;
        }
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) acc[mt][nt] = mfma16<BF16>(af[mt], bf[nt], acc[mt][nt]);
    }
    // D layout: col = lane & 15, row = 4 * (lane >> 4) + r
#pragma unroll
    for (int mt = 0; mt < TM; ++mt)
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int c = n0 + nt * 16 + li;
            if (c >= p.n) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + mt * 16 + 4 * kg + r;
                if (row < p.m) Z[(int64_t)row * p.n + c] = acc[mt][nt][r];
            }
        }
}

// Development switches that scripts and tests flip INSIDE one process ($DGA_B16_PLAN, _DEEP, _NO_TABLE, _WSK, _WSK_ODD) are looked up
// per call only when $DGA_B16_DEV is set (read once): a product process never calls getenv on the launch path (a few hundred
// nanoseconds on a launch of a few microseconds) and never races a setenv from another thread (tests/conftest.py sets it).
static const char *b16_dev_env(const char *name)
{
    static const bool dev = [] { const char *e = std::getenv("DGA_B16_DEV"); return e && *e && *e != '0'; }();
    return dev ? std::getenv(name) : nullptr;
}


// Tile and split-K of the tiled path (shared by the workspace size and the launch).  The largest tile that still gives
// every CU a workgroup; when even the shortest tile leaves most CUs idle (the reference's benchmark list,
// framework/benchmark/benchmark.py:24-44, is mostly M = 8..128 against N, K in the thousands: the y stream is the cost),
// K is cut so that every CU pulls on that stream, with fp32 slabs combined by a second kernel (as in the fp8 operator).
// raster group: a near-square patch of the tiles an XCD runs at the same time (one per CU here: 32 -> 4 rows x 8 columns)
static int b16_raster(int tiles_m)
{
    static const int forced = [] { const char *e = std::getenv("DGA_B16_RASTER"); return e ? std::atoi(e) : 0; }();
    if (forced > 0) return forced;
    return tiles_m >= 4 ? 4 : (tiles_m >= 2 ? 2 : 1);
}

struct B16Plan { int bm, bn, splitk, ks_per_split, tail, w8; };   // tail: the last partial round of a 256 x 256 / 128 x 256 raster in
                                                                   // sub-tiles of `tail` x 128 (0: none); w8: the 128 x 128 tile's
                                                                   // 8-wave three-stage build (one workgroup per CU)

// Swept plans of the operator's decode rows (the 16-bit counterpart of tuned/mi355x.csv; the reference keeps such winners in its
// CSV tiling cache, op_host/op_tiling/cache.cpp:22-101): for the (N, K) of the cold decode sweep (scripts/op16_plan_cold.py ->
// profiles/r04_op16_plan_cold.txt) the plan that won in each row bucket (0,8] (8,16] (16,32] (32,64] (64,128], where it beat the
// rule below by more than 6 % (the sweep times the rule first, which costs it ~3 %).  bm = 0: the one-launch workgroup split-K.  Generated by scripts/build_b16_plan_table.py.
struct B16Swept { int m_hi, n, k; short bm, bn, s; };
static const B16Swept kB16Swept[] = {     // the operator: y stored [N, K]
#include "b16_plans_mi355x.inc"
    {0, 0, 0, 0, 0, 0}};
static const B16Swept kB16SweptNN[] = {   // run_mmad_rtc / run_mmad_bench: y [K, N] read where it lies (profiles/r04_mmad_plan_cold.txt)
#include "b16_plans_mi355x_nn.inc"
    {0, 0, 0, 0, 0, 0}};
static const B16Swept *b16_swept(int batch, int m, int n, int k, bool nn)
{
    if (batch != 1 || m > 128) return nullptr;
    if (const char *e = b16_dev_env("DGA_B16_NO_TABLE"); e && *e == '1') return nullptr;   // (per call: the sweep flips it)
    const int m_hi = m <= 8 ? 8 : (m <= 16 ? 16 : (m <= 32 ? 32 : (m <= 64 ? 64 : 128)));
    for (const B16Swept *r = nn ? kB16SweptNN : kB16Swept; r->m_hi; ++r)
        if (r->m_hi == m_hi && r->n == n && r->k == k) return r;
    return nullptr;
}

// Tiles of at most 64 rows exist with two LDS stages (48 KB at most: up to three workgroups share a CU) and with FOUR (one workgroup per
// CU, three stages in flight): where the launch gives every CU at most one workgroup the deep build keeps three times the bytes in
// flight -- cold, bf16: 64 x 24576 x 1536 28.3 -> 22.1 us, 64 x 4096 x 7168 split 4 32.2 -> 22.3, 48 x 7168 x 4608 split 4 24.4 -> 18.8,
// 64 x 32768 x 512 13.1 -> 10.9; with more workgroups than CUs the shared CU wins (64 x 24576 x 1536 split 4: 32.4 against 34.4).
// $DGA_B16_DEEP = 0 / 1 overrides (development).
static bool b16_deep(const B16Plan &pl, int batch, int m, int n)
{
    if (pl.bm > 64) return false;
    if (const char *e = b16_dev_env("DGA_B16_DEEP")) return std::atoi(e) != 0;
    const int64_t items = static_cast<int64_t>(batch) * ((m + pl.bm - 1) / pl.bm) * ((n + pl.bn - 1) / pl.bn) * pl.splitk;
    return items <= device_cus();
}

static B16Plan b16_plan(int batch, int m, int n, int k, bool nn)
{
    const int ks_n = (k + 63) / 64;
    auto tiles_of = [&](int bm, int bn) { return static_cast<int64_t>(batch) * ((m + bm - 1) / bm) * ((n + bn - 1) / bn); };
    B16Plan pl{128, 128, 1, ks_n, 0, 0};
    auto split = [&](int s) {
        if (s > 1) { pl.ks_per_split = (ks_n + s - 1) / s; pl.splitk = (ks_n + pl.ks_per_split - 1) / pl.ks_per_split; }
    };
    if (const char *e = b16_dev_env("DGA_B16_PLAN")) {   // development: "bm,bn,splitk" (scripts/op16_plan_ab.py)
        int bm = 0, bn = 0, s = 1, tail = 0, w8 = 0;
        if (std::sscanf(e, "%d,%d,%d,%d,%d", &bm, &bn, &s, &tail, &w8) >= 2 && bm > 0 && bn > 0) {
            pl.bm = bm; pl.bn = bn; pl.w8 = (w8 && bm == 128 && bn == 128) ? 1 : 0;
            split(s);
            if (tail == 1) tail = bm / 2;   // quarter tiles
            pl.tail = (pl.splitk == 1 && batch == 1 && bn == 256 && (bm == 256 || bm == 128) && tail < bm &&
                       (tail == 128 || tail == 64 || tail == 32)) ? tail : 0;
            return pl;
        }
    }
    if (const B16Swept *r = b16_swept(batch, m, n, k, nn); r && r->bm) { pl.bm = r->bm; pl.bn = r->bn; split(r->s); return pl; }
    const int64_t cus = device_cus();            // 256 on a whole MI355X; fewer under a compute-partition mode
    // At most 128 rows over a long weight stream: the matmul is a read of y, and what reads fastest is the 128 x 256 tile with its
    // three 48 KB stages in flight per CU -- one workgroup per CU, split-K filling the CUs once with at least 8 k steps a slice
    // (cold sweep, profiles/r04_op16_plan_cold.txt: 129280 x 7168 at 32 / 64 rows 430 / 440 -> 310 us against the 128 x 128 tile
    // the fill rule below picks, 57344 x 8192 237 -> 171, 7168 x 18432 62 / 68 -> 54 / 56, 18432 x 7168 at 64 rows 63 -> 58).
    // Shorter streams stay with the rules below (within 2-8 % of the best plan there).
    // At most 128 rows and a raster of small tiles that fits the CUs once: the deep (four-stage) build of the smallest tile that covers
    // the rows (or half of them), split-K filling the CUs once with at least 12 k steps a slice -- every CU streams with three
    // stages in flight.  Taken where it fills at least three quarters of the CUs; fitted to the cold sweeps of both paths
    // (profiles/r04_op16_plan_cold.txt, r04_mmad_plan_cold.txt: mean distance from the best plan 10.3 % / 7.8 % -> 5.1 % / 2.9 %).
    // (two tile rows -- more than 64 rows -- read y twice through the L2: only where the stream is short; longer ones take the
    //  128 x 256 rule below: 128 x 8192 x 28672 149 -> 101 us)
    if (batch == 1 && (m <= 64 || (m <= 128 && static_cast<int64_t>(n) * k < (48ll << 20)))) {
        const int bm0 = m <= 16 ? 16 : (m <= 32 ? 32 : 64);
        int best_bm = 0, best_s = 1, best_fill = -1;
        for (int bm = bm0; bm >= 16 && bm >= bm0 / 2; bm /= 2) {
            const int64_t t = tiles_of(bm, 128);
            if (t > cus) continue;
            const int smax = static_cast<int>(std::min<int64_t>({cus / t, std::max(1, ks_n / 12), 12}));
            int s = 1;
            for (int c : {2, 3, 4, 6, 8, 12}) if (c <= smax) s = c;
            const int fill = static_cast<int>(100 * t * s / cus);   // per cent of the CUs
            if (fill > best_fill) { best_fill = fill; best_bm = bm; best_s = s; }
        }
        if (best_fill >= 75) { pl.bm = best_bm; pl.bn = 128; split(best_s); return pl; }
    }
    const int64_t nk = static_cast<int64_t>(n) * k;
    if (batch == 1 && m <= 128 && (nk >= (48ll << 20) || (m <= 64 && n >= 16384))) {
        const int64_t t = tiles_of(128, 256);
        pl.bm = 128; pl.bn = 256;
        split(static_cast<int>(std::max<int64_t>(1, std::min<int64_t>({t <= cus ? cus / t : 1, ks_n / 8, 12}))));
        return pl;
    }
    // More than one short tile row: (tile, split-K) by rounds x k steps x the tile's measured time per 64-wide k step, + the
    // combine of a split (device-timed A/B of the plans on nine mid-size shapes, scripts/op16_plan_ab.py ->
    // profiles/r03_op16_plan_ab.txt: 1024x4096x7168 84 -> 64 us, 1536x6144x4096 101 -> 78, 512x7168x4096 49 -> 41,
    // 1024x18432x7168 326 -> 287 against the fill-the-chip-with-the-biggest-tile rule below, which stays for M <= 64).
    if (m > 64) {
        // (the 128 x 128 tile twice: 4 waves and two LDS stages -- two workgroups share a CU -- and 8 waves with three stages, one
        //  workgroup per CU, 0.48 us a step where the 4-wave build alone on a CU takes 0.73: 1024 x 4096 x 7168 63.0 -> 57.2 us,
        //  512 x 7168 x 4096 41.1 -> 32.6, level with the vendor library's 58.2 / 34.0; scripts/op16_tail_ab.py)
        struct Cand { int bm, bn; double us_per_step; int wpc, w8; };
        static const Cand kCands[] = {{256, 256, 1.45, 1, 0}, {128, 256, 0.87, 1, 0}, {128, 128, 0.73, 2, 0}, {64, 128, 0.58, 2, 0},
                                      {128, 128, 0.48, 1, 1}};
        double best = 1e300;
        for (const Cand &c : kCands)
            for (int s : {1, 2, 3, 4, 6, 8, 12, 16}) {
                if (s > 1 && ks_n / s < 8) continue;
                if (s > 1 && static_cast<int64_t>(s) * batch * m * n * 4 > (256ll << 20)) continue;   // fp32 slabs <= 256 MiB (as the fp8 selector)
                const int per = (ks_n + s - 1) / s, s_eff = (ks_n + per - 1) / per;
                const int64_t items = tiles_of(c.bm, c.bn) * s_eff;
                const double rounds = std::ceil(static_cast<double>(items) / static_cast<double>(cus * c.wpc));
                const double share = static_cast<double>(std::min<int64_t>(c.wpc, (items + cus - 1) / cus));
                // a SINGLE round that leaves CUs idle runs its k steps faster than the full chip's figure (clock and memory headroom:
                // 168 tiles of 256 x 256 take 1.20 us a step, 140 take 1.11, 256 take 1.45 -- scripts/op16_plan_mid.py; the partial
                // last round of a longer raster does not: 350 tiles take two full rounds); the 64-row tile's figure is its
                // two-stage build's -- a launch of at most one workgroup per CU runs the deep build (b16_deep): 0.36
                const double busy = rounds > 1.0 ? 1.0 : std::max(0.5, std::min(1.0, static_cast<double>(items) / static_cast<double>(cus)));
                const double us_step = (c.bm == 64 && items <= cus) ? 0.36 : c.us_per_step;
                double t = 3.0 + rounds * per * us_step * std::pow(share, 0.6) * (0.55 + 0.45 * busy);
                if (s_eff > 1) t += 4.0 + static_cast<double>(s_eff) * batch * m * n * 8.0 / 5.0e6;
                if (t < best) { best = t; pl.bm = c.bm; pl.bn = c.bn; pl.splitk = s_eff; pl.ks_per_split = per; pl.tail = 0; pl.w8 = c.w8; }
            }
        // the whole rounds of a 256 x 256 / 128 x 256 raster as they are, the last partial round in sub-tiles (128 / 64 / 32 rows x
        // 128 columns, a second launch): 4-16 times as many CUs busy for a fraction of a round -- what the reference's Stream-K
        // handler is for (select_kernel.cpp:303-331) without partial sums: every output is one accumulation in k order, the bytes
        // are the single launch's (tests/test_op16_tail_gpu.py).  bf16, warm (scripts/op16_tail_ab.py -> profiles/r04_op16_tail_ab.txt):
        // 1024 x 18432 x 7168 289 -> 265 us, 5119 x 6997 x 9901 762 -> 705, 4608 x 4096 x 7168 260 -> 227, 2304 x 8192 x 4096 153 -> 137
        if (batch == 1) {
            static const Cand kSub[] = {{128, 128, 0.73, 2, 0}, {64, 128, 0.58, 2, 0}, {32, 128, 0.50, 2, 0}};
            for (int ci = 0; ci < 2; ++ci) {
                const Cand &c = kCands[ci];
                const int64_t tiles = tiles_of(c.bm, c.bn), tail = tiles % cus;
                if (tiles <= cus || tail == 0) continue;
                for (const Cand &q : kSub) {
                    if (q.bm >= c.bm || q.bm * 8 < c.bm) continue;
                    const int64_t items = tail * (c.bm / q.bm) * (c.bn / q.bn);
                    const double rounds = std::ceil(static_cast<double>(items) / static_cast<double>(cus * q.wpc));
                    const double share = static_cast<double>(std::min<int64_t>(q.wpc, (items + cus - 1) / cus));
                    const bool q8 = q.bm == 128 && items <= cus;   // (the sub-launch takes the 8-wave 128 x 128 build then)
                    const double t = 3.0 + static_cast<double>(tiles / cus) * ks_n * c.us_per_step + 2.0 +
                                     (q8 ? ks_n * 0.48 : rounds * ks_n * q.us_per_step * std::pow(share, 0.6));
                    if (t < best) { best = t; pl.bm = c.bm; pl.bn = c.bn; pl.splitk = 1; pl.ks_per_split = ks_n; pl.tail = q.bm; pl.w8 = 0; }
                }
            }
        }
        return pl;
    }
    const int64_t fill = cus * 3 / 4;            // "the big tile fills the chip": three quarters of the CUs
    if (tiles_of(256, 256) >= fill) { pl.bm = 256; pl.bn = 256; return pl; }
    if (tiles_of(128, 256) >= fill) { pl.bm = 128; pl.bn = 256; return pl; }   // 8 waves, three LDS stages
    if (tiles_of(128, 128) < fill && m <= 64) pl.bm = m > 32 ? 64 : (m > 16 ? 32 : 16);
    const int64_t tiles = tiles_of(pl.bm, pl.bn);
    if (tiles * 4 <= cus * 3 && ks_n >= 16) {
        int s = static_cast<int>(std::min<int64_t>({2 * cus / tiles, ks_n / 8, 16}));
        // slab write + read stays below half of the operand read
        while (s > 1 && static_cast<int64_t>(s) * batch * m * n * 8 * 2 > static_cast<int64_t>(batch) * (m + n) * k * 2) --s;
        if (s > 1) {
            pl.ks_per_split = (ks_n + s - 1) / s;
            pl.splitk = (ks_n + pl.ks_per_split - 1) / pl.ks_per_split;
        }
    }
    return pl;
}

static size_t b16_workspace_bytes(int batch, int m, int n, int k, const void *x)
{
    if (batch <= 0 || m <= 0 || n <= 0 || k <= 0) return 0;
    const size_t kp = (static_cast<size_t>(k) + 63) / 64 * 64;
    size_t bytes = ((static_cast<size_t>(batch) * n * kp * 2 + 255) & ~size_t(255));       // yT
    const bool x_in_place = (k % 64 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (!x_in_place) bytes += ((static_cast<size_t>(batch) * m * kp * 2 + 255) & ~size_t(255));  // padded x
    const B16Plan pl = b16_plan(batch, m, n, k, true);
    if (pl.splitk > 1) bytes += ((static_cast<size_t>(pl.splitk) * batch * m * n * 4 + 255) & ~size_t(255));  // fp32 slabs
    return bytes + 256;
}

template <class Cfg, bool BF16, int PP = 0, bool NN = false, bool OUT16 = false>
static int launch_tiled(const B16Params &p, int batch, hipStream_t stream)
{
    auto kfn = gemm_b16_nt_f32_kernel<Cfg, BF16, PP, NN, OUT16>;
    constexpr int lds = (PP == 2 ? 2 : Cfg::STAGES) * (Cfg::A_BYTES + Cfg::B_BYTES);
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    if (record_hip(attr_err[dev]) != DGA_OK) return DGA_E_HIP;
    const unsigned grid = p.launch_tiles > 0 ? static_cast<unsigned>(p.launch_tiles)
                                             : static_cast<unsigned>(batch) * p.tiles_m * p.tiles_n * (p.splitk > 1 ? p.splitk : 1);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), lds, stream, p);
    return record_hip(hipGetLastError());
}

static int launch_b16(const void *x, const void *y, float *z, int batch, int m, int n, int k, int dtype,
                      void *workspace, size_t workspace_bytes, hipStream_t stream)
{
    if (batch < 0 || m < 0 || n < 0 || k < 0) return DGA_E_SHAPE;
    if (dtype != DGA_DT_BF16 && dtype != DGA_DT_FP16) return DGA_E_DTYPE;
    if (batch == 0 || m == 0 || n == 0) return DGA_OK;
    if (!z || ((!x || !y) && k != 0)) return DGA_E_NULL;
    const size_t need = b16_workspace_bytes(batch, m, n, k, x);
    if (workspace && workspace_bytes < need) return DGA_E_WORKSPACE;
    const size_t kp = (static_cast<size_t>(k) + 63) / 64 * 64;
    // y read where it lies ([K][N], transposing LDS reads) when the 16-byte DMA chunks line up; otherwise y is first
    // transposed into the workspace (the re-layout the reference does on the way into L1, generate_code.hpp:250-260)
    static const int no_direct = [] { const char *e = std::getenv("DGA_B16_TRANSPOSE"); return e ? std::atoi(e) : 0; }();
    const bool direct = !no_direct && k > 0 && (k % 64 == 0) && (n % 8 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(y) & 15) == 0) && (static_cast<int64_t>(k) * n * 2 < 0x7FFFFFFFll) &&
                        (static_cast<int64_t>(k) * 257 * 2 < 0x7FFFFFFFll);
    if ((workspace || direct) && k > 0 && kp * 257 * 2 < 0x7FFFFFFFull) {
        uint8_t *ws = static_cast<uint8_t *>(workspace);
        uint16_t *yt = reinterpret_cast<uint16_t *>(ws);
        const size_t yt_bytes = (static_cast<size_t>(batch) * n * kp * 2 + 255) & ~size_t(255);
        const bool x_in_place = (k % 64 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
        const uint16_t *xs = static_cast<const uint16_t *>(x);
        if (!direct) {
            dim3 tg((n + 63) / 64, static_cast<unsigned>(kp / 64), batch);
            const int y_vec = (n % 8 == 0) && ((reinterpret_cast<uintptr_t>(y) & 15) == 0);
            hipLaunchKernelGGL(transpose_b16_kernel, tg, dim3(256), 0, stream, static_cast<const uint16_t *>(y), yt, k, n,
                               static_cast<int>(kp), static_cast<int64_t>(k) * n, static_cast<int64_t>(n) * kp, y_vec);
            if (!x_in_place) {
                uint16_t *xp = reinterpret_cast<uint16_t *>(ws + yt_bytes);
                const int64_t rows = static_cast<int64_t>(batch) * m;
                if (pad_rows(xs, xp, rows, nullptr, nullptr, 0, static_cast<int64_t>(k) * 2, static_cast<int64_t>(kp) * 2, stream) != DGA_OK)
                    return DGA_E_HIP;
                xs = xp;
            }
            if (record_hip(hipGetLastError()) != DGA_OK) return DGA_E_HIP;
        }
        B16Params p{};
        p.x = xs; p.z = z;
        p.yt = direct ? static_cast<const uint16_t *>(y) : yt;
        p.m = m; p.n = n; p.k = static_cast<int>(kp);
        p.ldx = x_in_place ? k : static_cast<int64_t>(kp);
        p.ldy = direct ? static_cast<int64_t>(n) : static_cast<int64_t>(kp);
        p.x_bs = static_cast<int64_t>(m) * p.ldx;
        p.y_bs = direct ? static_cast<int64_t>(k) * n : static_cast<int64_t>(n) * kp;
        p.z_bs = static_cast<int64_t>(m) * n;
        static const int plain = [] { const char *e = std::getenv("DGA_B16_PLAIN"); return e ? std::atoi(e) : 0; }();
        const bool bf = dtype == DGA_DT_BF16;
        B16Plan pl = b16_plan(batch, m, n, k, true);
        if (!workspace) { pl.splitk = 1; pl.ks_per_split = static_cast<int>(kp / 64); }  // no room for the slabs
        p.batch = batch;
        p.splitk = pl.splitk;
        p.ks_per_split = pl.ks_per_split;
        size_t slab_at = yt_bytes;
        if (!x_in_place) slab_at += (static_cast<size_t>(batch) * m * kp * 2 + 255) & ~size_t(255);
        p.partial = pl.splitk > 1 ? reinterpret_cast<float *>(ws + slab_at) : nullptr;
        auto go = [&](auto cfg, auto pp) -> int {
            using Cfg = decltype(cfg);
            constexpr int PPv = decltype(pp)::value;
            if (!p.tail_sub) {
                p.tiles_m = (m + Cfg::kBM - 1) / Cfg::kBM;
                p.tiles_n = (n + Cfg::kBN - 1) / Cfg::kBN;
                p.raster_group = b16_raster(p.tiles_m);
            }
            if (direct)
                return bf ? launch_tiled<Cfg, true, PPv, true>(p, batch, stream) : launch_tiled<Cfg, false, PPv, true>(p, batch, stream);
            return bf ? launch_tiled<Cfg, true, PPv>(p, batch, stream) : launch_tiled<Cfg, false, PPv>(p, batch, stream);
        };
        using P0 = std::integral_constant<int, 0>;
        using P2 = std::integral_constant<int, 2>;
        const bool deep = b16_deep(pl, batch, m, n);
        // main launch over the whole rounds, then the parent raster's remaining tiles as sub-tiles (pl.tail)
        auto go_tail = [&](auto cfg, auto pp) -> int {
            using Cfg = decltype(cfg);
            const int tiles = ((m + Cfg::kBM - 1) / Cfg::kBM) * ((n + Cfg::kBN - 1) / Cfg::kBN), cus = static_cast<int>(device_cus());
            const int tail = tiles % cus, main_tiles = tiles - tail;
            if (!pl.tail || tail == 0 || main_tiles == 0) return go(cfg, pp);
            p.launch_tiles = main_tiles;
            if (int rc = go(cfg, pp)) return rc;
            const int sm = Cfg::kBM / pl.tail, sn = Cfg::kBN / 128;
            p.launch_tiles = tail * sm * sn; p.tail_begin = main_tiles; p.tail_sub = sm | (sn << 8);
            if (pl.tail == 128) return p.launch_tiles <= cus ? go(GemmCfg<128, 128, 2, 4, 3>{}, P0{}) : go(GemmCfg<128, 128, 2, 2>{}, P0{});
            if (pl.tail == 64) return go(GemmCfg<64, 128, 1, 4>{}, P0{});
            return go(GemmCfg<32, 128, 1, 4>{}, P0{});
        };
        int rc;
        if (pl.bm == 256) rc = plain ? go(GemmCfg<256, 256, 4, 2>{}, P0{}) : go_tail(GemmCfg<256, 256, 4, 2>{}, P2{});
        else if (pl.bm == 128 && pl.bn == 256) rc = go_tail(GemmCfg<128, 256, 2, 4, 3>{}, P0{});
        else if (pl.bm == 128) rc = pl.w8 ? go(GemmCfg<128, 128, 2, 4, 3>{}, P0{}) : go(GemmCfg<128, 128, 2, 2>{}, P0{});
        else if (pl.bm == 64) rc = deep ? go(GemmCfg<64, 128, 1, 4, 4>{}, P0{}) : go(GemmCfg<64, 128, 1, 4>{}, P0{});
        else if (pl.bm == 32) rc = deep ? go(GemmCfg<32, 128, 1, 4, 4>{}, P0{}) : go(GemmCfg<32, 128, 1, 4>{}, P0{});
        else rc = deep ? go(GemmCfg<16, 128, 1, 4, 4>{}, P0{}) : go(GemmCfg<16, 128, 1, 4>{}, P0{});
        if (rc != DGA_OK || pl.splitk <= 1) return rc;
        const int64_t total = static_cast<int64_t>(batch) * m * n;
        hipLaunchKernelGGL(splitk_reduce_f32_kernel, dim3(static_cast<unsigned>((total / 4 + 255) / 256 + 1)), dim3(256), 0,
                           stream, p.partial, z, total, pl.splitk);
        return record_hip(hipGetLastError());
    }
    // no workspace: fragments gathered straight from global memory (correct, slow)
    B16DirectParams p{};
    p.x = static_cast<const uint16_t *>(x);
    p.y = static_cast<const uint16_t *>(y);
    p.z = z;
    p.m = m; p.n = n; p.k = k;
    p.x_bs = static_cast<int64_t>(m) * k;
    p.y_bs = static_cast<int64_t>(k) * n;
    p.z_bs = static_cast<int64_t>(m) * n;
    dim3 grid((n + 127) / 128, (m + 127) / 128, batch);
    if (dtype == DGA_DT_BF16)
        hipLaunchKernelGGL(mmad_nn_f32_kernel<true>, grid, dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL(mmad_nn_f32_kernel<false>, grid, dim3(256), 0, stream, p);
    return record_hip(hipGetLastError());
}

// ---- decode rows of the operator: the one-launch workgroup split-K (gemm_b16_wsk_kernel.hpp) ------------------------------
template <bool BF16, int TN, int D, int TM = 1>
static int launch_b16_wsk_one(const B16Params &p, unsigned grid, hipStream_t stream)
{
    auto kfn = gemm_b16_wsk_kernel<BF16, TN, D, TM>;
    constexpr int kLds = 8 * (D * (16 * TM + TN * 16) * 128 + 16 * TM * TN * 16 * 4);
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kLds, stream, p);
    return record_hip(hipGetLastError());
}

// M <= 32, operands 16-byte aligned with rows of whole 128-byte k steps (in place or padded); DGA_E_TILING: not a launch it takes
template <bool BF16>
static int launch_b16_wsk(const B16Params &p_in, hipStream_t stream)
{
    B16Params p = p_in;
    p.ks_per_split = (p.k / 64 + 7) / 8;
    if (const char *e = b16_dev_env("DGA_B16_WSK_ODD"); e && *e == '1' && p.ks_per_split % 2 == 0) ++p.ks_per_split;
    if (p.m > 32 || p.m <= 0 || p.batch != 1 || (p.k % 64) || !p.z16 || ((p.ldx * 2) & 15) || ((p.ldy * 2) & 15) ||
        (reinterpret_cast<uintptr_t>(p.x) & 15) || (reinterpret_cast<uintptr_t>(p.yt) & 15) ||
        static_cast<int64_t>(p.m) * p.ldx * 2 >= 0x7FFFFFFFll)
        return DGA_E_TILING;
    const int nt = (p.n + 15) / 16;
    const int64_t cus = device_cus();
    const unsigned g = static_cast<unsigned>(nt < cus ? nt : cus);
    const int per = static_cast<int>((nt + g - 1) / g), p2 = (per + 1) / 2, p3 = (per + 2) / 3;
    if (p.m > 16)   // two 16-row tiles of x: the LDS holds 2 n-tiles x 2 stages or 1 x 3 (160 KB with the slab either way)
        return per <= 1 ? launch_b16_wsk_one<BF16, 1, 3, 2>(p, g, stream) : launch_b16_wsk_one<BF16, 2, 2, 2>(p, g, stream);
    if (per <= 1) return launch_b16_wsk_one<BF16, 1, 4>(p, g, stream);
    if (per == 3 || (per >= 5 && p3 < p2 && per % 3 != 1)) return launch_b16_wsk_one<BF16, 3, 2>(p, g, stream);
    return launch_b16_wsk_one<BF16, 2, 3>(p, g, stream);
}

// ---- the aclnn operator in its own dtypes: out[M,N] (16-bit) = self[M,K] . mat2, mat2 stored [N,K] (NT) -----------------
static size_t b16_nt_workspace_bytes(int m, int n, int k, const void *a, const void *b)
{
    if (m <= 0 || n <= 0 || k <= 0) return 0;
    const size_t kp = (static_cast<size_t>(k) + 63) / 64 * 64;
    const bool in_place = (k % 64 == 0) && (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0);
    size_t bytes = 0;
    if (!in_place) bytes += ((static_cast<size_t>(m) * kp * 2 + 255) & ~size_t(255)) + ((static_cast<size_t>(n) * kp * 2 + 255) & ~size_t(255));
    const B16Plan pl = b16_plan(1, m, n, k, false);
    if (pl.splitk > 1) bytes += (static_cast<size_t>(pl.splitk) * m * n * 4 + 255) & ~size_t(255);
    return bytes ? bytes + 256 : 0;
}

static int launch_b16_nt(const void *a, const void *b, void *out, int m, int n, int k, int dtype, void *workspace,
                         size_t workspace_bytes, hipStream_t stream)
{
    if (m < 0 || n < 0 || k < 0) return DGA_E_SHAPE;
    if (dtype != DGA_DT_BF16 && dtype != DGA_DT_FP16) return DGA_E_DTYPE;
    if (m == 0 || n == 0) return DGA_OK;
    if (!out || ((!a || !b) && k != 0)) return DGA_E_NULL;
    if (workspace && workspace_bytes < b16_nt_workspace_bytes(m, n, k, a, b)) return DGA_E_WORKSPACE;
    const bool bf = dtype == DGA_DT_BF16;
    const size_t kp = (static_cast<size_t>(k) + 63) / 64 * 64;
    const bool in_place = (k % 64 == 0) && (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0);
    if (k == 0 || (!in_place && !workspace) || kp * 257 * 2 >= 0x7FFFFFFFull) {
        // nothing to pad into (or K = 0: zeros): element-wise kernel
        dim3 grid((n + 15) / 16, (m + 15) / 16);
        if (bf) hipLaunchKernelGGL(gemm_b16_nt_generic_kernel<true>, grid, dim3(256), 0, stream, static_cast<const uint16_t *>(a),
                                   static_cast<const uint16_t *>(b), static_cast<uint16_t *>(out), m, n, k);
        else hipLaunchKernelGGL(gemm_b16_nt_generic_kernel<false>, grid, dim3(256), 0, stream, static_cast<const uint16_t *>(a),
                                static_cast<const uint16_t *>(b), static_cast<uint16_t *>(out), m, n, k);
        return record_hip(hipGetLastError());
    }
    uint8_t *ws = static_cast<uint8_t *>(workspace);
    size_t at = 0;
    const uint16_t *as = static_cast<const uint16_t *>(a), *bs = static_cast<const uint16_t *>(b);
    if (!in_place) {  // rows zero-padded to whole 128-byte k steps (the PaddingCommon variant's role)
        uint16_t *ap = reinterpret_cast<uint16_t *>(ws);
        at = (static_cast<size_t>(m) * kp * 2 + 255) & ~size_t(255);
        uint16_t *bp = reinterpret_cast<uint16_t *>(ws + at);
        at += (static_cast<size_t>(n) * kp * 2 + 255) & ~size_t(255);
        if (pad_rows(as, ap, m, bs, bp, n, static_cast<int64_t>(k) * 2, static_cast<int64_t>(kp) * 2, stream) != DGA_OK) return DGA_E_HIP;
        as = ap; bs = bp;
    }
    B16Params p{};
    p.x = as; p.yt = bs; p.z = nullptr; p.z16 = static_cast<uint16_t *>(out);
    p.m = m; p.n = n; p.k = static_cast<int>(kp);
    p.ldx = in_place ? k : static_cast<int64_t>(kp);
    p.ldy = p.ldx;
    p.x_bs = static_cast<int64_t>(m) * p.ldx;
    p.y_bs = static_cast<int64_t>(n) * p.ldy;
    p.z_bs = static_cast<int64_t>(m) * n;
    p.batch = 1;
    // decode rows: one launch, the K slices are the waves of a workgroup (gemm_b16_wsk_kernel.hpp).  When it was built the tile path
    // was two-stage tiles by the fill rule and it won on 51 of 57 cold decode shapes by 15-35 % (profiles/r04_op16_wsk_cold.txt);
    // the deep small tiles took most of that back.  $DGA_B16_WSK = 0 / 1 overrides the rule.
    const char *wsk_e = b16_dev_env("DGA_B16_WSK");   // (read per call, like $DGA_B16_PLAN: the tests flip it inside one process)
    const int wsk_env = wsk_e ? std::atoi(wsk_e) : -1;
    // The rule, against the best tile plan of the cold sweep (profiles/r04_op16_plan_cold.txt; since the small tiles have their deep
    // builds): at <= 16 rows the one launch is ahead by 5-25 % on streams of N K <= 32 M elements and level (+-5 %) on longer ones
    // unless the workgroups' reads of x weigh in (N < 8192: 5120 x 13824 +12..22 %, 7168 x 18432 +10 %); K = 512 gives a wave one k
    // step (+28..39 %); rows a multiple of 32 KB apart collide in the memory channels.  17..32 rows (two 16-row tiles of x, a
    // shallower ring): ahead only on short streams of short rows -- K <= 4096 and N K <= 16 M (4096 x 4096 0.97 of the best tile
    // plan, 7168 x 2048 0.95, 7168 x 1536 0.86, 1024 x 4096 0.90) -- and 5-40 % behind elsewhere.
    const B16Swept *swept = m <= 32 ? b16_swept(1, m, n, k, false) : nullptr;
    const int64_t nk = static_cast<int64_t>(n) * kp;
    const bool wsk_rule = kp >= 1024 && (kp % 16384) != 0 &&
                          (m <= 16 ? (nk <= (32ll << 20) || n >= 8192) : (m <= 32 && kp <= 4096 && nk <= (16ll << 20)));
    if (wsk_env >= 0 ? wsk_env != 0 : (swept ? swept->bm == 0 : wsk_rule)) {
        const int rc = bf ? launch_b16_wsk<true>(p, stream) : launch_b16_wsk<false>(p, stream);
        if (rc != DGA_E_TILING) return rc;
    }
    B16Plan pl = b16_plan(1, m, n, k, false);
    if (!workspace) { pl.splitk = 1; pl.ks_per_split = static_cast<int>(kp / 64); }
    p.splitk = pl.splitk;
    p.ks_per_split = pl.ks_per_split;
    p.partial = pl.splitk > 1 ? reinterpret_cast<float *>(ws + at) : nullptr;
    auto go = [&](auto cfg, auto pp) -> int {
        using Cfg = decltype(cfg);
        constexpr int PPv = decltype(pp)::value;
        if (!p.tail_sub) {
            p.tiles_m = (m + Cfg::kBM - 1) / Cfg::kBM;
            p.tiles_n = (n + Cfg::kBN - 1) / Cfg::kBN;
            p.raster_group = b16_raster(p.tiles_m);
        }
        return bf ? launch_tiled<Cfg, true, PPv, false, true>(p, 1, stream) : launch_tiled<Cfg, false, PPv, false, true>(p, 1, stream);
    };
    using P0 = std::integral_constant<int, 0>;
    using P2 = std::integral_constant<int, 2>;
    const bool deep = b16_deep(pl, 1, m, n);
    auto go_tail = [&](auto cfg, auto pp) -> int {
        using Cfg = decltype(cfg);
        const int tiles = ((m + Cfg::kBM - 1) / Cfg::kBM) * ((n + Cfg::kBN - 1) / Cfg::kBN), cus = static_cast<int>(device_cus());
        const int tail = tiles % cus, main_tiles = tiles - tail;
        if (!pl.tail || tail == 0 || main_tiles == 0) return go(cfg, pp);
        p.launch_tiles = main_tiles;
        if (int rc = go(cfg, pp)) return rc;
        const int sm = Cfg::kBM / pl.tail, sn = Cfg::kBN / 128;
        p.launch_tiles = tail * sm * sn; p.tail_begin = main_tiles; p.tail_sub = sm | (sn << 8);
        if (pl.tail == 128) return p.launch_tiles <= cus ? go(GemmCfg<128, 128, 2, 4, 3>{}, P0{}) : go(GemmCfg<128, 128, 2, 2>{}, P0{});
        if (pl.tail == 64) return go(GemmCfg<64, 128, 1, 4>{}, P0{});
        return go(GemmCfg<32, 128, 1, 4>{}, P0{});
    };
    int rc;
    static const int w4_env = [] { const char *e = std::getenv("DGA_B16_W4"); return e ? std::atoi(e) : 0; }();
    if (pl.bm == 256 && w4_env && !pl.tail && pl.splitk == 1) {
        p.tiles_m = (m + 255) / 256; p.tiles_n = (n + 255) / 256; p.raster_group = b16_raster(p.tiles_m);
        rc = launch_b16_w4(p, bf, stream);
    }
    else if (pl.bm == 256) rc = go_tail(GemmCfg<256, 256, 4, 2>{}, P2{});
    else if (pl.bm == 128 && pl.bn == 256) rc = go_tail(GemmCfg<128, 256, 2, 4, 3>{}, P0{});
    else if (pl.bm == 128) rc = pl.w8 ? go(GemmCfg<128, 128, 2, 4, 3>{}, P0{}) : go(GemmCfg<128, 128, 2, 2>{}, P0{});
    else if (pl.bm == 64) rc = deep ? go(GemmCfg<64, 128, 1, 4, 4>{}, P0{}) : go(GemmCfg<64, 128, 1, 4>{}, P0{});
    else if (pl.bm == 32) rc = deep ? go(GemmCfg<32, 128, 1, 4, 4>{}, P0{}) : go(GemmCfg<32, 128, 1, 4>{}, P0{});
    else rc = deep ? go(GemmCfg<16, 128, 1, 4, 4>{}, P0{}) : go(GemmCfg<16, 128, 1, 4>{}, P0{});
    if (rc != DGA_OK || pl.splitk <= 1) return rc;
    const int64_t total = static_cast<int64_t>(m) * n;
    if (bf) hipLaunchKernelGGL(splitk_reduce_16_kernel<true>, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream,
                               p.partial, p.z16, total, pl.splitk);
    else hipLaunchKernelGGL(splitk_reduce_16_kernel<false>, dim3(static_cast<unsigned>((total + 255) / 256)), dim3(256), 0, stream,
                            p.partial, p.z16, total, pl.splitk);
    return record_hip(hipGetLastError());
}

}  // namespace dga

extern "C" {

size_t dga_catlass_dynamic_matmul_workspace_bytes(int m, int n, int k, const void *self, const void *mat2)
{
    return dga::b16_nt_workspace_bytes(m, n, k, self, mat2);
}

int dga_catlass_dynamic_matmul(const void *self, const void *mat2, void *out, int m, int n, int k, int dtype,
                               void *workspace, size_t workspace_bytes, void *stream)
{
    return dga::launch_b16_nt(self, mat2, out, m, n, k, dtype, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

size_t dga_mmad_workspace_bytes(int batch, int m, int n, int k, const void *x)
{
    return dga::b16_workspace_bytes(batch, m, n, k, x);
}

int dga_run_mmad_rtc_ws(const void *x, const void *y, float *z, int batch, int m, int n, int k, int dtype,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    return dga::launch_b16(x, y, z, batch, m, n, k, dtype, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int dga_run_mmad_rtc(const void *x, const void *y, float *z, int batch, int m, int n, int k, int dtype, void *stream)
{
    return dga::launch_b16(x, y, z, batch, m, n, k, dtype, nullptr, 0, static_cast<hipStream_t>(stream));
}

static int check_bench_params(const int32_t *params_host, int m, int n, int k)
{
    // The 28 ints steer the Ascend kernel's L1/L0 blocking only; they are validated (the knobs must be positive,
    // as the reference's derivation divides by them) and otherwise have no CDNA4 meaning.
    if (params_host) {
        for (int i = 0; i < 6; ++i)
            if (params_host[i] <= 0) return DGA_E_RANGE;
        if (params_host[6] != m || params_host[7] != n || params_host[8] != k) return DGA_E_SHAPE;
    }
    return DGA_OK;
}

int dga_run_mmad_bench_ws(const void *x, const void *y, float *z, int m, int n, int k, int dtype,
                          const int32_t *params_host, void *workspace, size_t workspace_bytes, void *stream)
{
    const int rc = check_bench_params(params_host, m, n, k);
    if (rc != DGA_OK) return rc;
    return dga::launch_b16(x, y, z, 1, m, n, k, dtype, workspace, workspace_bytes, static_cast<hipStream_t>(stream));
}

int dga_run_mmad_bench(const void *x, const void *y, float *z, int m, int n, int k, int dtype,
                       const int32_t *params_host, void *stream)
{
    return dga_run_mmad_bench_ws(x, y, z, m, n, k, dtype, params_host, nullptr, 0, stream);
}

}  // extern "C"
