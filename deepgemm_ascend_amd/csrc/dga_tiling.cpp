// Host tiling / kernel selection for libdga_hip.so -- the CDNA4 retarget of
//   /root/reference/aclnn_catlass_dynamic_matmul/op_host/op_tiling/
//       utils.cpp        (CeilDiv, RoundUp, BalanceWorkload :28, JudgeSpace :59, GetMaxK1 :66)
//       do_tiling.cpp    (DoTilingLayout01 :58-111 -- the op is hard-wired NT,
//                         catlass_dynamic_matmul_tiling.cpp:83-84)
//       select_kernel.cpp (handler chain Small -> StreamK -> PaddingCommon -> Common :333-369)
//       cache.cpp / csv.cpp (TilingCache :22-101, CSV::Document :31-140)
// and of its Python mirror get_best_config/tiling_calculator.py.
//
// Two modes share one arithmetic core:
//   * platform.xcdNum <= 1 (Ascend numbers): the reference's own NT tile search, padding cost model
//     and handler chain replayed, 16-element granularity, L1/L0C limits -- exists so tests can pin
//     the restatement against the reference's golden tuples (tests/golden/op_tiling_vectors.json).
//   * platform.xcdNum  > 1 (MI355X): same search skeleton (start from the aspect-ratio tile,
//     balance the block count against the core count, bound by on-chip space), but the result
//     is drawn from the compiled kernel menu (dga_launch.hip kVariants), space is LDS + VGPR
//     accumulators, K step is one 128-wide scale block, and padding variants do not exist
//     (NZ re-layout is an Ascend artefact; the LDS image is swizzled by the DMA source address).
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <mutex>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "dga_hip.h"
#include "dga_internal.hpp"

namespace dga {
namespace tiling {

inline uint32_t ceil_div(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline uint32_t round_up(uint32_t v, uint32_t al) { return ceil_div(v, al) * al; }

// ---- shared with the reference: on-chip space test and the K-step ladder ----------------
// utils.cpp:59-64: double-buffered A and B tiles must fit "L1", the fp32 tile must fit "L0C".
bool judge_space(uint32_t m1, uint32_t n1, uint32_t k1, const dga_platform_t &pf, uint32_t data_size)
{
    const uint64_t staged = 2ull * data_size * (static_cast<uint64_t>(m1) * k1 + static_cast<uint64_t>(k1) * n1);
    return staged <= pf.l1Size && 4ull * m1 * n1 <= pf.l0CSize;
}

// utils.cpp:66-77: largest of {1024,512,256,128} that fits, else 512/dataSize.
uint32_t max_k1(uint32_t m1, uint32_t n1, const dga_platform_t &pf, uint32_t data_size)
{
    for (uint32_t k1 : {1024u, 512u, 256u, 128u})
        if (judge_space(m1, n1, k1, pf, data_size)) return k1;
    return 512 / data_size;
}

// utils.cpp:28-40: shrink m1 in steps of 16 while the block count does not exceed the
// core-count-rounded count of the starting tile; clamp tiles to the (16-rounded) problem.
void balance_workload(uint32_t m, uint32_t n, uint32_t &m1, uint32_t &n1, uint32_t threshold,
                      const dga_platform_t &pf)
{
    const uint32_t cap = round_up(ceil_div(m, m1) * ceil_div(n, n1), pf.coreNum);
    while (m1 > threshold && ceil_div(m, m1 - 16) * ceil_div(n, n1) <= cap) m1 -= 16;
    if (m < m1) m1 = round_up(m, 16);
    if (n < n1) n1 = round_up(n, 16);
}

// do_tiling.cpp:58-111 (A row-major, B column-major).  Bandwidth is layout-neutral in NT, so the
// search only balances work: pick 128x256 or 256x128 by aspect ratio, balance both axes, then grow
// the short axis while space allows and a whole round of cores is saved.
void do_tiling_nt_reference(dga_tiling_t &t, const dga_platform_t &pf, uint32_t data_size)
{
    const uint32_t m = t.m, n = t.n, k = t.k;
    uint32_t m1 = 128, n1 = 256, k1 = 256;
    // NOTE: the reference evaluates m*k + k*n and m*n in uint32 (do_tiling.cpp:66) -- kept.
    const double ratio = static_cast<double>(m * k + k * n) / (m * n);
    const bool tall = m > n && (ratio > 0.1 || n < 256);
    if (tall) {
        m1 = 256; n1 = 128;
        balance_workload(m, n, m1, n1, 64, pf);
        balance_workload(n, m, n1, m1, 64, pf);
    } else {
        balance_workload(n, m, n1, m1, 64, pf);
        balance_workload(m, n, m1, n1, 64, pf);
    }
    const uint32_t cap = round_up(ceil_div(m, m1) * ceil_div(n, n1), pf.coreNum);
    if (m < n) {
        for (uint32_t cand = n1; judge_space(m1, cand + 16, k1, pf, data_size);) {
            cand += 16;
            if (ceil_div(m, m1) * ceil_div(n, cand) <= cap - pf.coreNum) n1 = cand;
        }
        balance_workload(m, n, m1, n1, 64, pf);
        balance_workload(n, m, n1, m1, 64, pf);
    } else {
        for (uint32_t cand = m1; judge_space(cand + 16, n1, k1, pf, data_size);) {
            cand += 16;
            if (ceil_div(m, cand) * ceil_div(n, n1) <= cap - pf.coreNum) m1 = cand;
        }
        balance_workload(n, m, n1, m1, 64, pf);
        balance_workload(m, n, m1, n1, 64, pf);
    }
    if (k >= 65536) {
        const bool wide = m < n || (ratio < 0.1 && n >= 256);
        m1 = wide ? 128 : 256;
        n1 = wide ? 256 : 128;
    }
    k1 = max_k1(m1, n1, pf, data_size);
    t.m1 = static_cast<uint16_t>(m1); t.n1 = static_cast<uint16_t>(n1); t.k1 = static_cast<uint16_t>(k1);
}

// ---- padding cost model, reference mode only (select_kernel.cpp:22-80, 82-268) ----------------
// Prices the Ascend ND->NZ re-layout of an operand on the vector cores against loading it
// unpadded.  It has no CDNA4 meaning (the LDS image is swizzled by the DMA source address) and is
// restated only so that the handler chain can be replayed against the reference's golden tuples.
// GetBandwidth (:22-80): GB/s a cube core reaches loading `rows` x `cols` pieces of a matrix whose
// contiguous axis is `src_cols` long.
double cube_bandwidth(uint32_t rows, uint32_t cols, uint64_t src_cols)
{
    const double d = cols;
    // 6th-order fit of unaligned-load bandwidth vs piece width (constants: select_kernel.cpp:23-29)
    double bw = 0.000000000000020146121020 * std::pow(d, 6) - 0.000000000012456944162142 * std::pow(d, 5) -
                0.000000006738536427145036 * std::pow(d, 4) + 0.000007301215580838747961 * std::pow(d, 3) -
                0.002146456956750821074703 * std::pow(d, 2) + 0.312849910814454512664184 * d + 0.1;
    if (cols == src_cols && cols <= 128 && cols % 16 == 0) bw = 60;
    if (src_cols >= 65536) bw = 1;
    if (src_cols % 256 == 0) bw *= 100.0 / 30;
    else if (src_cols % 128 == 0) bw *= 80.0 / 30;
    else if (src_cols % 64 == 0) bw *= 50.0 / 30;
    else if (src_cols % 16 == 0) bw *= 40.0 / 30;
    bw = std::min(bw, 80.0);
    const double r = rows;
    if (cols % 256 == 0) {
        if (rows < 16) bw *= -0.003332381309698882569659 * r * r + 0.113578920178116271610946 * r + 0.016102868630357251855667;
    } else if (cols % 32 == 0) {
        if (rows < 32) bw *= -0.000298086120946179481978 * r * r + 0.045309519479127147167929 * r + 0.035130178145161221336945;
    } else if (rows < 64) {
        bw *= 0.000001809180573350345869 * r * r * r - 0.000469676727179688081274 * r * r +
              0.038963259596073690493867 * r + 0.003942641759904389614499;
    }
    return bw;
}

struct OperandCost {
    uint64_t outer, inner;       // matrix extent across / along the contiguous axis
    double bw_plain, bw_padded;  // cube-core GB/s without / with NZ padding
    double bw_vec;               // vector-core GB/s for the padding pass
    uint64_t cube_bytes;         // bytes the busiest cube core loads
    uint64_t vec_bytes;          // bytes the busiest vector core re-lays out
    uint32_t vec_tasks;
};

// one operand of GetPaddingTag: (outer, inner) extents, (tile_outer, tile_inner) piece, `tile_cnt` tiles along
// the non-K axis, `actual` = min(extent, tile) of that axis
OperandCost operand_cost(uint64_t outer, uint64_t inner, uint32_t piece_rows, uint32_t piece_cols, uint64_t elems,
                         uint32_t tile_cnt, uint32_t actual, uint32_t round_max, uint32_t block_dim, uint32_t k,
                         uint32_t k1, uint32_t splitk, const dga_platform_t &pf)
{
    OperandCost c{};
    c.outer = outer; c.inner = inner;
    c.bw_vec = (elems * 2 > 192ull * 1024 * 1024) ? 10 : 30;   // beyond the 192 MB L2 (:113-117)
    c.bw_plain = cube_bandwidth(piece_rows, piece_cols, inner);
    if (tile_cnt < block_dim / 2 && k <= k1 && tile_cnt <= 2) c.bw_plain = c.bw_plain / (block_dim / tile_cnt) * 1.5;
    c.bw_padded = 80;
    if (piece_rows < 16) c.bw_padded *= static_cast<double>(piece_rows) / 16;
    c.cube_bytes = static_cast<uint64_t>(round_max) * actual * ceil_div(k, splitk) * 2;
    // padding simulator (:147-180): 16-row tasks of up to 48 KB spread over 2 vector cores per cube core
    uint32_t task_rows = 16, task_cols = 48 * 1024 / 2 / 16;
    if (inner < task_cols) task_cols = static_cast<uint32_t>(inner);
    if (outer < task_rows) task_rows = static_cast<uint32_t>(outer);
    task_cols = round_up(static_cast<uint32_t>(inner) / ceil_div(static_cast<uint32_t>(inner), task_cols), 16);
    c.vec_tasks = ceil_div(static_cast<uint32_t>(outer), task_rows) * ceil_div(static_cast<uint32_t>(inner), task_cols);
    c.vec_bytes = static_cast<uint64_t>(ceil_div(c.vec_tasks, pf.coreNum * 2)) * task_cols * task_rows * 2;
    return c;
}

// GetPaddingTag (:82-268), NT layout: A is [m][k], B is [n][k] (both K-contiguous)
void padding_tags_reference(dga_tiling_t &t, const dga_platform_t &pf)
{
    const uint32_t m = t.m, n = t.n, k = t.k, m1 = t.m1, n1 = t.n1, k1 = t.k1, sk = t.splitkFactor;
    const uint32_t tiles_m = ceil_div(m, m1), tiles_n = ceil_div(n, n1);
    const uint32_t tasks = tiles_m * tiles_n * sk;
    const uint32_t block_dim = std::min(tasks, pf.coreNum);
    const uint32_t round_max = ceil_div(tasks, pf.coreNum);
    const OperandCost A = operand_cost(m, k, std::min(m, m1), std::min(k, k1), static_cast<uint64_t>(m) * k, tiles_m,
                                       std::min(m, m1), round_max, block_dim, k, k1, sk, pf);
    const OperandCost B = operand_cost(n, k, std::min(n, n1), std::min(k, k1), static_cast<uint64_t>(k) * n, tiles_n,
                                       std::min(n, n1), round_max, block_dim, k, k1, sk, pf);
    const double head = sk > 1 ? 1.0 : 1 + 7 * static_cast<double>(block_dim) / pf.coreNum;  // us
    auto us = [](uint64_t bytes, double gbps) { return static_cast<double>(bytes) / gbps / 1000; };
    const double t00 = us(A.cube_bytes, A.bw_plain) + us(B.cube_bytes, B.bw_plain);
    const double t01 = us(A.cube_bytes, A.bw_plain) + us(B.cube_bytes, B.bw_padded) + us(B.vec_bytes, B.bw_vec) + head;
    const double t10 = us(A.cube_bytes, A.bw_padded) + us(B.cube_bytes, B.bw_plain) + us(A.vec_bytes, A.bw_vec) + head;
    const double t11 = us(A.cube_bytes, A.bw_padded) + us(B.cube_bytes, B.bw_padded) + us(A.vec_bytes, A.bw_vec) +
                       us(B.vec_bytes, B.bw_vec) + head + 2;
    uint8_t pa = DGA_PADDING_NONE, pb = DGA_PADDING_NONE;
    double best = t00;
    if (t01 < best) { best = t01; pa = DGA_PADDING_NONE; pb = DGA_PADDING_NZ; }
    if (t10 < best) { best = t10; pa = DGA_PADDING_NZ; pb = DGA_PADDING_NONE; }
    if (t11 < best) { best = t11; pa = DGA_PADDING_NZ; pb = DGA_PADDING_NZ; }
    auto forced = [](const OperandCost &c) {
        if ((c.inner < 8 || (c.inner < 32 && c.inner % 16 != 0)) && c.outer > 512) return true;
        return c.outer >= 2048 && c.inner > 8192 && c.inner % 8192 == 0;  // "meta conflicts" (:229-235)
    };
    if (forced(A)) pa = DGA_PADDING_NZ;
    if (forced(B)) pb = DGA_PADDING_NZ;
    uint8_t pc = DGA_PADDING_NONE;
    if (static_cast<uint64_t>(m) * n > 2048ull * 2048 && n > 256 && n % 128 != 0) {
        const uint64_t total = static_cast<uint64_t>(m) * k * tiles_n * 2 + static_cast<uint64_t>(k) * n * tiles_m * 2 +
                               static_cast<uint64_t>(m) * n * 2;
        if (total < 192ull * 1024 * 1024) pc = DGA_PADDING_ND;
    }
    t.paddingTagA = pa; t.paddingTagB = pb; t.paddingTagC = pc;
    uint32_t vec_a = (pa && A.inner > 192) ? A.vec_tasks : 0, vec_b = (pb && B.inner > 192) ? B.vec_tasks : 0;
    const uint32_t vec = std::max(vec_a, vec_b);
    const uint32_t bd_vec = std::min(ceil_div(vec, 2), pf.coreNum);
    t.blockDim = (pa || pb) ? std::max(block_dim, bd_vec) : block_dim;
}

// ---- handler chain, reference mode (select_kernel.cpp:270-369): Small -> StreamK -> PaddingCommon -> Common
void select_reference(dga_tiling_t &t, const dga_platform_t &pf)
{
    do_tiling_nt_reference(t, pf, 2);
    // Small (:278-293)
    padding_tags_reference(t, pf);
    if (!t.paddingTagA && !t.paddingTagB && !t.paddingTagC) {
        const uint32_t blocks = ceil_div(t.m, t.m1) * ceil_div(t.n, t.n1);
        if (blocks <= pf.coreNum && t.k <= t.k1) {
            t.kernelSerial = DGA_KERNEL_SMALL;
            return;
        }
    }
    // Stream-K (:303-331): best-bandwidth tile, all cores, when the tail round is < 80 % full and K is long
    {
        const uint32_t sb = ceil_div(t.m, 128) * ceil_div(t.n, 256);
        const uint32_t rem = sb % pf.coreNum;
        if (sb > pf.coreNum && sb < 8 * pf.coreNum && rem > 0 && rem < 0.8 * pf.coreNum && t.k > 3072) {
            t.m1 = 128; t.n1 = 256; t.k1 = 256;
            padding_tags_reference(t, pf);
            t.blockDim = pf.coreNum;
            t.kernelSerial = DGA_KERNEL_STREAMK;
            return;
        }
    }
    if (t.paddingTagA || t.paddingTagB || t.paddingTagC) {  // PaddingCommon (:295-301)
        t.kernelSerial = DGA_KERNEL_PADDING_COMMON;
        return;
    }
    t.blockDim = std::min(ceil_div(t.m, t.m1) * ceil_div(t.n, t.n1), pf.coreNum);  // Common (:270-276)
    t.kernelSerial = DGA_KERNEL_COMMON;
}

// ---- MI355X mode -------------------------------------------------------------------------
// Cost of running the problem with workgroup tile (bm, bn): rounds of the chip x time of one
// tile, where one tile is bound by the slower of its MFMA work and its operand streaming.
struct MenuEntry { int bm, bn, wm, wn, lds; };

static std::vector<MenuEntry> menu()
{
    std::vector<MenuEntry> v;
    for (int i = 0; i < variant_count(); ++i) {
        MenuEntry e{};
        variant_info(i, &e.bm, &e.bn, &e.wm, &e.wn, &e.lds);
        v.push_back(e);
    }
    return v;
}

// MFMA issue efficiency of a wave tile: each k block costs (TM*TN) MFMAs of 32 cycles and
// (TM + TN) fragment reads of 2 ds_read_b128 (4 LDS cycles each, shared by the CU's waves);
// small wave tiles are LDS-read bound.
static double tile_cycles_per_kblock(const MenuEntry &e)
{
    const int waves = e.wm * e.wn;
    const double tm = e.bm / e.wm / 16.0, tn = e.bn / e.wn / 16.0;
    const double waves_per_simd = std::max(1.0, waves / 4.0);
    const double mfma = tm * tn * 32.0 * waves_per_simd;
    const double lds = (tm + tn) * 2.0 * 4.0 * waves;         // LDS array cycles, all waves
    const double dma = (e.bm + e.bn) * 128.0 / 64.0;          // ~64 B/clk/CU from L2 into LDS
    return std::max({mfma, lds, dma}) + 200.0;                // + barrier/issue overhead per k block
}

}  // namespace tiling

// Tail in quarter tiles (kernelSerial 5): more than one wave of 256x256 tiles with a small remainder -- the remainder would
// cost a whole extra round at 1/8..1/4 occupancy; covered by 128x128 tiles (second launch) it occupies four times the
// CUs for about half a round.  Applied to a dense tiling whatever chose the tile (heuristic or predictor).
// LDS bytes of the compiled build a tiling resolves to: exact (tile, stages, wave grid) match, else the menu's first
// build of that tile and stage count, else of that tile
static uint32_t menu_lds_bytes(const dga_tiling_t &t)
{
    const int stages = t.stages == 3 ? 3 : 2;
    for (int pass = 0; pass < 3; ++pass)
        for (int i = 0; i < variant_count(); ++i) {
            int bm, bn, wm, wn, lds;
            variant_info(i, &bm, &bn, &wm, &wn, &lds);
            if (bm != t.m1 || bn != t.n1) continue;
            if (pass < 2 && variant_stages(i) != stages) continue;
            if (pass == 0 && t.wavesM && (wm != t.wavesM || wn != t.wavesN)) continue;
            return static_cast<uint32_t>(lds);
        }
    return t.ldsBytes;
}

// Loader waves (dispatchPolicyTag 4): every 4-wave, 3-stage tile has a build with four extra waves that carry the whole
// LDS-DMA.  Measured on one box (scripts/lc_ab.py): 128x256 60.7 -> 55.7 us at 4096x2048x7168 (against the 8-wave 2x4
// build), 128x128 40.4 -> 36.0 us at 2048x2048x7168 and 37.4 -> 31.0 us at 1024x2048x7168, 64x256 37.9 -> 32.2 us; the
// masked grouped stream -2 %.  The output bytes are those of the plain loop, so whatever chose the tile (heuristic,
// predictor, a swept row from before these builds existed) is upgraded here.
//
// Persistent form (dispatchPolicyTag 5, gemm_fp8_persistent_kernel.hpp): where the raster holds more tiles than the chip
// has CUs, one workgroup per CU walks its share and the ring runs across tile boundaries.  Same bits again.  Measured
// (scripts/ps_check.py): 256 x (128, 7168, 2048) full mask 770 -> 746 us, random mask 672 -> 657, decode masks 592 -> 577;
// dense 128x256 rasters of 2-4 tiles per CU -3..-10 %.  Split-K and the quarter-tile tail keep the one-tile builds.
// upgrade_plain = false: a swept row whose file carries the dispatchPolicyTag column names the build the sweep TIMED (the plain
// loop and the loader-wave build are separate candidates there), so policy 0 stays policy 0; only the persistent forms --
// folded into their one-tile siblings' records by the sweep -- are still chosen by rule.
void prefer_loader_waves(dga_tiling_t &t, bool upgrade_plain)
{
    // (not the 128x256 tile under split-K: its plain 3-stage build has 8 computing waves, and with the short k range of a split
    //  the 4 + 4 loader-wave build is 6-11 % slower -- device-timed sweep of 641 shapes, profiles/r03_predictor; every other tile
    //  is the same kernel within 0.1 % under split-K)
    if (upgrade_plain && t.dispatchPolicyTag == DGA_POLICY_PLAIN && t.stages == 3 &&
        !(t.splitkFactor > 1 && t.m1 == 128 && t.n1 == 256)) {
        for (int i = 0; i < variant_count(); ++i) {
            int bm, bn, wm, wn, lds;
            variant_info(i, &bm, &bn, &wm, &wn, &lds);
            if (bm != t.m1 || bn != t.n1 || variant_stages(i) != 3 || !variant_has_loader_waves(i)) continue;
            t.wavesM = static_cast<uint8_t>(wm); t.wavesN = static_cast<uint8_t>(wn);
            t.ldsBytes = static_cast<uint32_t>(lds);
            t.dispatchPolicyTag = DGA_POLICY_LOADER_WAVES;
            break;
        }
    }
    if (t.dispatchPolicyTag == DGA_POLICY_LOADER_WAVES && t.splitkFactor <= 1 && t.kernelSerial != DGA_KERNEL_STREAMK_TAIL &&
        t.m1 && t.n1 && !(t.contiguous && t.m1 > DGA_CONTIGUOUS_M_ALIGNMENT)) {
        const uint64_t groups = t.contiguous ? 1 : std::max<uint32_t>(1, t.groups);
        const uint64_t tiles = groups * ((t.m + t.m1 - 1) / t.m1) * ((t.n + t.n1 - 1) / t.n1);
        // the weight stream of the grouped layouts runs the persistent build whatever its tile count: that kernel moves the
        // weights and the outputs with the non-temporal policy (dga_launch.hip), worth more than the tile boundaries
        const bool weight_stream = t.contiguous ? (t.groups > 1 && static_cast<uint64_t>(t.m) <= static_cast<uint64_t>(t.groups) * DGA_CONTIGUOUS_M_ALIGNMENT)
                                                : t.groups > 1;
        // dense rasters: the persistent builds of the tall tiles win (128x256 -3..-9 %, 128x128 -5..-9 %, 64x256 -1..-6 %), those of
        // the short ones lose (64x128 +9..+20 %, 16x128 +33 %: the tile list walk and the loaders' look-ahead cost more than a tile
        // boundary of theirs does) -- the same device-timed sweep; round 2's rule (every tile) came from launch intervals timed
        // through Python, which could not tell kernels under ~12 us apart
        const bool tall = (t.m1 == 128 && (t.n1 == 256 || t.n1 == 128)) || (t.m1 == 64 && t.n1 == 256);
        // (... unless the whole problem is a few tiles of a few k blocks: 16 groups x 64 rows x (1536, 4096) -- 192 tiles of 32 k
        //  blocks, 20 us of work -- takes 31 us on the persistent build and 20-22 on the one-tile ones; at 192 tiles of 64 k
        //  blocks the persistent build is ahead again, 93.6 against 100)
        const uint64_t kb = (static_cast<uint64_t>(t.k) + 127) / 128;
        if ((tiles > device_cus() && tall) || (weight_stream && tiles * kb >= 8192)) t.dispatchPolicyTag = DGA_POLICY_PERSISTENT;
    }
    // ... and the continuous 256x256 kernel has its own persistent form (dispatchPolicyTag 6,
    // gemm_fp8_cont_persistent_kernel.hpp) for dense rasters of full tiles: the next tile's first stages are fetched from
    // inside the last k blocks.  Worth 3-6 % where a tile is short (K <= 4096) and a CU runs more than one
    // (scripts/cps_check.py: 4096x7168x2048 61.7 -> 58.1 us, 4096x8192x2048 67.6 -> 63.7, 8192x4096x4096 117.2 -> 114.5;
    // 8192^3 unchanged: the dispatcher already starts a CU's next workgroup while the last one's stores drain).
    if (t.dispatchPolicyTag == DGA_POLICY_CONTINUOUS && t.m1 == 256 && t.n1 == 256 && t.splitkFactor <= 1 && !t.contiguous &&
        std::max<uint32_t>(1, t.groups) == 1 && t.m % 256 == 0 && t.n % 256 == 0 && t.k % 128 == 0 && t.k >= 256) {
        const uint64_t tiles = static_cast<uint64_t>(t.m / 256) * (t.n / 256);
        if (tiles > device_cus()) t.dispatchPolicyTag = DGA_POLICY_CONTINUOUS_PERSISTENT;
    }
}

void apply_tail_split(dga_tiling_t &t, uint32_t cus)
{
    if (t.splitkFactor > 1 || t.m1 != 256 || t.n1 != 256 || t.groups > 1 || t.contiguous || !cus) return;
    const uint64_t blocks = static_cast<uint64_t>((t.m + 255) / 256) * ((t.n + 255) / 256);
    if (blocks <= cus) return;
    const uint32_t tail = static_cast<uint32_t>(blocks % cus);
    if (tail == 0 || tail * 4 > cus) return;
    t.kernelSerial = DGA_KERNEL_STREAMK_TAIL;
    t.blockDim = static_cast<uint32_t>(blocks - tail) + tail * 4;
}

namespace tiling {

// ---- dense problems: (tile, split-K) by a cost model fitted to a device-timed sweep -----------------------------------------
// 641 + 120 shapes x every (tile, split-K, stages, policy) candidate, two runs merged by min, each candidate timed by graph replay (harness/sweep.py
// graph_us; the launch intervals of rounds 1-2 were taken through Python and could not tell kernels under ~12 us apart,
// profiles/r03_host_overhead.txt), short-M shapes on operand sets rotated past the Infinity Cache.  scripts/fit_heuristic.py fits
//   T = launch + rounds x (k blocks per item x us_per_kblock[tile] x share^e + prologue)            (the tile pass)
//       floored by launch + bytes / min(HBM rate, workgroups in flight x per-workgroup stream rate)  (the operand stream;
//       a short-M weight stream is cold, so every tile row streams B again)
//       + combine + slab bytes / slab rate                                                            (split-K only)
// in log time on the 641 training shapes (rmse 0.13); picking by it costs 2.1 % over the best candidate on the 120 HELD-OUT shapes
// (geomean of pick / best 1.021, p90 1.10, max 1.23; 1.013 / 1.26 in sample) where the tile-first rule it replaces cost 21-24 %
// (max 3.5x: mid-M, short-N, long-K shapes that want tall tiles and a split, which a rule that fills the chip with tiles first
// never considers).  profiles/r03_predictor/heuristic_fit.txt.
struct DenseTileCost { int bm, bn; double us_per_kblock; };
struct DenseCostModel {
    const DenseTileCost *tiles; int n_tiles;
    double share_exponent, launch_us, prologue_us, combine_us, slab_bytes_per_us, hbm_bytes_per_us, workgroup_bytes_per_us;
};
static const DenseTileCost kDenseTileCost[] = {{256, 256, 1.547}, {128, 256, 0.933}, {256, 128, 1.153}, {128, 128, 0.579},
                                               {64, 256, 0.650},  {64, 128, 0.367},  {32, 256, 0.449},  {32, 128, 0.256},
                                               {16, 256, 0.394},  {16, 128, 0.213}};
static const DenseCostModel kFastModel = {kDenseTileCost, 10, 0.8845, 2.66, 1.31, 3.53, 5.09e6, 8.0e6, 48.6e3};
// the bf16-exact policy's menu (dga_launch_menu_e.hip: three stages, no loader waves), fitted the same way on its own
// device-timed sweep (scripts/bx_sweep.py, 214 + 120 shapes): held-out pick / best 1.019 geomean, max 1.28 (the hand-set model
// of this policy's first version, fitted to 19 shapes timed through Python: 1.069, max 1.47)
// (refitted after the MFMA results moved to VGPRs -- the Makefile's -amdgpu-mfma-vgpr-form note: the 128x128 / 64x256 / 64x128 / 32x128
//  builds lost a v_accvgpr_read per MFMA and 14-19 % of their time on mid-M shapes, profiles/r03_vgpr_form.txt)
static const DenseTileCost kBf16xTileCost[] = {{128, 256, 1.583}, {128, 128, 0.966}, {64, 256, 1.075}, {64, 128, 0.677}, {32, 128, 0.605}};
static constexpr double kBf16x32RowTallUs = 0.442;   // the 32-row build on a problem of more than 256 rows (warm operands): its own figure
static const DenseCostModel kBf16xModel = {kBf16xTileCost, 5, 0.529, 1.57, 3.02, 2.98, 5.10e6, 5.35e6, 39.9e3};
static constexpr uint32_t kDenseSplits[] = {1, 2, 3, 4, 5, 6, 8, 16};   // what the sweeps cover
// fp32 split-K slabs a tiling may ask for: the caller keeps one grow-only workspace per stream (api._scratch), so this bounds
// the scratch a serving process pins per stream through tiling choices alone (+ the odd-K padding copies of its operands)
static constexpr uint64_t kMaxSlabBytes = 256ull << 20;
static bool tile_has_three_stages(int bm, int bn)
{
    for (int i = 0; i < variant_count(); ++i) {
        int vm, vn, wm, wn, lds;
        variant_info(i, &vm, &vn, &wm, &wn, &lds);
        if (vm == bm && vn == bn && variant_stages(i) == 3) return true;
    }
    return false;
}

// one LDS stage of a tile (dga_device_common.hpp GemmCfg with 256 DMA threads)
static uint32_t dense_stage_bytes(uint32_t bm, uint32_t bn) { return std::max(bm, 32u) * 128 + bn * 128 + ((bm + 8 + 255) / 256) * 1024; }

static double dense_cost_us(const DenseCostModel &mo, uint32_t m, uint32_t n, uint32_t k, const DenseTileCost &c, uint32_t splitk,
                            uint32_t stages, uint32_t cus, uint32_t lds_per_cu, uint32_t *splitk_eff)
{
    const uint32_t kb = ceil_div(std::max(k, 1u), 128), per = ceil_div(kb, splitk), s = ceil_div(kb, per);
    *splitk_eff = s;
    const uint64_t tiles_m = ceil_div(m, c.bm), tiles = tiles_m * ceil_div(n, c.bn), items = tiles * s;
    const uint64_t wpc = std::max<uint64_t>(1, std::min<uint64_t>(lds_per_cu / (stages * dense_stage_bytes(c.bm, c.bn)), 4));
    double rounds = std::ceil(static_cast<double>(items) / static_cast<double>(cus * wpc));
    if (&mo == &kFastModel && c.bm == 256 && c.bn == 256 && s == 1 && tiles > cus) {   // a small last wave is cut along K (apply_tail_split)
        const uint64_t tail = tiles % cus;
        if (tail > 0 && tail * 4 <= cus) rounds = static_cast<double>(tiles / cus) + 0.5;
    }
    if (&mo == &kBf16xModel && c.bm == 128 && c.bn == 256 && s == 1 && tiles > cus) {
        // the bf16-exact tile's last partial round in 64 x 128 quarter tiles (dga_launch.hip, kernelSerial 5): a quarter tile alone on a CU
        // takes 0.6 of a parent tile's time, two to a CU 0.85 (scripts/bx_tail_ab.py, profiles/r05_bx_tail_ab.txt)
        const uint64_t tail = tiles % cus;
        if (tail > 0 && tail * 2 <= cus) rounds = static_cast<double>(tiles / cus) + (tail * 4 <= cus ? 0.62 : 0.85);
    }
    const double share = static_cast<double>(std::min<uint64_t>(wpc, (items + cus - 1) / cus));
    double t = mo.launch_us + rounds * (per * c.us_per_kblock * std::pow(share, mo.share_exponent) + mo.prologue_us);
    const double bytes = static_cast<double>(m) * k + static_cast<double>(n) * k * (m <= 256 ? static_cast<double>(tiles_m) : 1.0) +
                         2.0 * m * n;
    const double in_flight = static_cast<double>(std::min<uint64_t>(items, cus * wpc));
    t = std::max(t, mo.launch_us + bytes / std::min(mo.hbm_bytes_per_us, in_flight * mo.workgroup_bytes_per_us));
    if (s > 1) t += mo.combine_us + static_cast<double>(s) * m * n * 8.0 / mo.slab_bytes_per_us;
    return t;
}

void select_mi355x(dga_tiling_t &t, const dga_platform_t &pf, uint32_t groups, uint32_t expected_m,
                   bool contiguous = false)
{
    // Grouped masked-M: the tile height covers m_max (the rows allocated per group), not expected_m.  masked_m[g] may
    // exceed the hint, and every extra tile row of a group streams that group's whole B again, while rows beyond
    // masked_m cost nothing but skipped waves: at G = 256 x (128, 7168, 2048) with 0..32 rows per group the 128-row tile
    // streams 5.7 TB/s where a 16-row tile (the hint's choice) reaches 2.8 (scripts/grouped_decode_perf.py).
    (void)expected_m;
    const uint32_t m_eff = t.m;
    const uint32_t kb = ceil_div(std::max(t.k, 1u), 128);
    double best = 1e300;
    MenuEntry pick{};
    bool found = false;
    std::vector<MenuEntry> seen;
    for (const MenuEntry &e : menu()) {
        bool dup = false;
        for (const MenuEntry &q : seen) dup |= (q.bm == e.bm && q.bn == e.bn);
        if (dup) continue;  // later entries of a tile size are alternative builds (waves / stages), chosen below
        seen.push_back(e);
        if (static_cast<uint64_t>(e.lds) > pf.l1Size) continue;
        if (4ull * e.bm * e.bn > pf.l0CSize) continue;  // accumulators: JudgeSpace's L0C test on VGPRs
        // contiguous-grouped layout: a tile must not straddle two group segments (aligned to 128 rows)
        // Long groups (>= 512 rows on average) take the 256x256 tile all the same: it runs a second pass on the tiles
        // that straddle two groups (about every second group boundary) and still wins by 15-25 % (scripts/contig_ab.py).
        if (contiguous) {
            // (... where its raster fills the CUs' rounds: 4 groups x 1024 rows x (2048, 4096) is 128 such tiles -- half the CUs --
            //  and takes 49 us against the 128 x 256 tile's 38; x (5120, 5120), 320 tiles = 1.25 rounds, 129 against 112;
            //  scripts/grouped_selector_regret.py)
            const uint64_t tiles256 = static_cast<uint64_t>(ceil_div(t.m, 256)) * ceil_div(t.n, 256);
            const double fill256 = static_cast<double>(tiles256) / (std::ceil(static_cast<double>(tiles256) / pf.coreNum) * pf.coreNum);
            const bool tall = t.m / std::max(1u, t.groups) >= 512 && t.n >= 256 && fill256 >= 0.8;
            if (tall ? !(e.bm == 2 * DGA_CONTIGUOUS_M_ALIGNMENT && e.bn == 256)
                     : (e.bm > DGA_CONTIGUOUS_M_ALIGNMENT || DGA_CONTIGUOUS_M_ALIGNMENT % e.bm))
                continue;
        }
        // M fits one tile row (decode / grouped shapes): B is the whole stream, so never cut M (every extra tile row
        // re-reads B) -- take the smallest tile height that covers M and let N tiles and split-K supply parallelism
        if (m_eff <= 128 && e.bm != static_cast<int>(std::max(16u, round_up(m_eff, m_eff <= 16 ? 16 : m_eff <= 32 ? 32 : m_eff <= 64 ? 64 : 128))))
            continue;
        const uint32_t wg_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(pf.l1Size / e.lds, 2048 / (e.wm * e.wn * 64)));
        const uint64_t tiles = static_cast<uint64_t>(groups) * ceil_div(m_eff, e.bm) * ceil_div(t.n, e.bn);
        const uint64_t slots = static_cast<uint64_t>(pf.coreNum) * wg_per_cu;
        double rounds = std::ceil(static_cast<double>(tiles) / slots);
        // 256x256: a small last partial wave is cut along K (apply_tail_split) and costs a fraction of a round
        if (groups == 1 && !contiguous && e.bm == 256 && e.bn == 256 && tiles > pf.coreNum) {
            const uint64_t tail = tiles % pf.coreNum;
            if (tail > 0 && tail * 4 <= pf.coreNum) rounds = static_cast<double>(tiles / pf.coreNum) + 0.5;
        }
        // co-resident workgroups share the CU's MFMA pipes
        const double share = std::min<double>(wg_per_cu, std::ceil(static_cast<double>(tiles) / pf.coreNum));
        double cost = rounds * share * kb * tile_cycles_per_kblock(e);
        // wasted rows of a partially filled tile are paid in full
        cost *= 1.0 + 1e-3 * (e.bm > m_eff ? (e.bm - m_eff) / 16.0 : 0.0);
        if (cost < best) { best = cost; pick = e; found = true; }
    }
    uint32_t dense_splitk = 0;   // > 0: (tile, split-K) chosen together by the fitted cost model
    if (groups == 1 && !contiguous) {
        double best_us = 1e300;
        for (const MenuEntry &e : seen) {
            if (static_cast<uint64_t>(e.lds) > pf.l1Size || 4ull * e.bm * e.bn > pf.l0CSize) continue;
            const DenseTileCost *c = nullptr;
            for (const DenseTileCost &q : kDenseTileCost)
                if (q.bm == e.bm && q.bn == e.bn) c = &q;
            if (!c) continue;
            if (c->bn == 256 && c->bm <= 32 && t.m > 256) continue;   // never the best of a tall problem (1 shape of 381), and their warm builds differ
            for (uint32_t sk : kDenseSplits) {
                if (sk > 1 && (kb < 4 * sk || static_cast<uint64_t>(sk) * t.m * t.n * 4 > kMaxSlabBytes)) continue;  // >= 4 k blocks per split, slabs <= 256 MiB
                uint32_t s_eff = 1;
                double us = dense_cost_us(kFastModel, t.m, t.n, t.k, *c, sk, tile_has_three_stages(c->bm, c->bn) ? 3 : 2, pf.coreNum,
                                          static_cast<uint32_t>(pf.l1Size), &s_eff);
                us *= 1.0 + 1e-3 * (e.bm > static_cast<int>(t.m) ? (e.bm - static_cast<int>(t.m)) / 16.0 : 0.0);   // ties: the tile with fewer idle rows
                if (us < best_us) { best_us = us; pick = e; dense_splitk = s_eff; found = true; }
            }
        }
    }
    if (!found) { t.m1 = t.n1 = 0; return; }
    t.m1 = static_cast<uint16_t>(pick.bm); t.n1 = static_cast<uint16_t>(pick.bn); t.k1 = 128;
    t.wavesM = static_cast<uint8_t>(pick.wm); t.wavesN = static_cast<uint8_t>(pick.wn);
    t.stages = 2;
    t.ldsBytes = pick.lds;
    // schedule: the 8-wave 256x256 tile runs the continuous pipeline (dispatchPolicyTag 2: -16 % cycles per k block,
    // +3..5 % wall at 4096^3); with one wave per SIMD (the 4-wave tiles) the plain loop is faster
    t.dispatchPolicyTag = (pick.bm == 256 && pick.bn == 256) ? 2 : 0;
    // The tiles whose stage is <= 48 KB have a three-stage build that keeps two refills in flight: +6 % GB/s on the
    // HBM-bound grouped stream, and the r01 sweep picked it for every shape on these tiles (profiles/r01_sweep).
    if ((pick.bm == 128 && pick.bn == 256 && pick.wm == 2 && pick.wn == 2) || (pick.bm == 128 && pick.bn == 128) ||
        (pick.bm == 64 && pick.bn == 256))
        t.stages = 3;
    // dense: the same sweep has the three-stage build (with its loader waves where it has them) ahead on every tile, cold by
    // 13-32 % and warm by 3-26 % (except the 256-wide short tiles on warm operands, which tall problems do not get)
    if (dense_splitk && tile_has_three_stages(pick.bm, pick.bn)) t.stages = 3;
    // The masked grouped stream on short tiles (at most 64 rows a group), off the tuned table: the three-stage build -- with its
    // loader waves in the persistent form where the tile has them (16 / 64 x 128: prefer_loader_waves below) -- instead of the
    // two-stage plain loop this function used to leave them with: 64 groups x 16 rows x (3072, 8192) 291 -> 260 us, x (1536, 4096)
    // 99 -> 85, 16 x 64 x (1536, 4096) 29 -> 20-22 (every legal build timed on 12 unseen grouped problems:
    // scripts/grouped_selector_regret.py -> profiles/r04_grouped_selector_regret.txt).  A 16-row group takes the 128-wide tile (the
    // one with loader waves: 272 against the 256-wide tile's 300-315 us at 64 x 16 x (5120, 5120)); the 32-row tiles have no such
    // build and are level at two and three stages: left alone.
    if (groups > 1 && !contiguous && pick.bm <= 64 && pick.bm != 32) {
        if (pick.bm == 16 && pick.bn == 256)
            for (const MenuEntry &e : seen)
                if (e.bm == 16 && e.bn == 128) { pick = e; t.n1 = 128; t.wavesM = static_cast<uint8_t>(e.wm); t.wavesN = static_cast<uint8_t>(e.wn); t.ldsBytes = e.lds; }
        if (tile_has_three_stages(pick.bm, pick.bn)) t.stages = 3;
    }
    // 128x256 with three stages has two builds: 4 waves (2x2) and 8 waves (2x4, two per SIMD).  The masked grouped
    // stream (HBM-bound) is 3 % faster on 4 waves, everything compute-bound 3-14 % faster on 8 (4096x2048x7168: 66.8 ->
    // 61.8 us; scripts/steady_ab.py, scripts/contig_ab.py).
    // (the contiguous layout with one 128-row block per group is the same HBM-bound stream as the masked layout: 4 waves)
    const bool weight_stream = groups > 1 || (contiguous && t.m / std::max(1u, t.groups) <= DGA_CONTIGUOUS_M_ALIGNMENT);
    if (pick.bm == 128 && pick.bn == 256 && t.stages == 3 && !weight_stream) { t.wavesM = 2; t.wavesN = 4; }
    // ... and on the weight stream the 4-wave build takes four extra loader waves (dispatchPolicyTag 4): the refill leaves
    // the computing waves' instruction streams, -2 % time on 256 x (128, 7168, 2048) (profiles/r02_grouped_ablation.txt)
    // (the weight stream keeps 2x2 computing waves; prefer_loader_waves() below adds the loader waves to every 3-stage pick)
    const uint32_t tiles_m = ceil_div(t.m, t.m1), tiles_n = ceil_div(t.n, t.n1);
    const uint64_t blocks = static_cast<uint64_t>(groups) * tiles_m * tiles_n;
    t.blockDim = static_cast<uint32_t>(blocks) * ((contiguous && t.m1 > DGA_CONTIGUOUS_M_ALIGNMENT) ? 2 : 1);
    t.paddingTagA = t.paddingTagB = t.paddingTagC = DGA_PADDING_NONE;
    // variant menu in the reference's order: Small (one tile per core, single K step) -> Stream-K -> Common.
    // Stream-K handler (select_kernel.cpp:303-331, CDNA4 reading): when the tiles fill less than half of the CUs and K
    // is long, K is cut so that every CU streams a share of the operands; the fp32 partial tiles are combined by a
    // second kernel.  Worth it only while the partial slabs stay small next to the operand stream.
    t.kernelSerial = (blocks <= pf.coreNum && t.k <= t.k1) ? DGA_KERNEL_SMALL : DGA_KERNEL_COMMON;
    // Decode rows whose raster fills at most half of the CUs with the split the model chose: the model's form misses the 100-128
    // tile range (N ~ 12-16 K at 128 columns a tile), where it sees no reason to split and a split of 2 is 15-20 % faster cold
    // (32 x 14336 x 4096 23.3 -> 18.7 us, 32 x 13824 x 5120 27.9 -> 22.7: profiles/r04_sweep_decode3/).  Double it while the launch
    // still fits the CUs once and a slice keeps >= 16 k blocks -- on 30 unseen (N, K) x 3 row counts it changes 12 picks, all
    // for the better (0.75-0.89 of the time, profiles/r04_sweep_decode3/fill_bump_ab.txt); with 8 k blocks a slice (K = 2048) the
    // same doubling costs 2-10 %.  $DGA_NO_FILL_BUMP = 1: the model's own split (A/B).
    static const bool no_bump = [] { const char *e = std::getenv("DGA_NO_FILL_BUMP"); return e && std::atoi(e) != 0; }();
    if (!no_bump && dense_splitk >= 1 && groups == 1 && !contiguous && t.m <= 128) {
        while (blocks * dense_splitk * 2 <= pf.coreNum && kb / (2 * dense_splitk) >= 16 &&
               static_cast<uint64_t>(2 * dense_splitk) * t.m * t.n * 4 <= kMaxSlabBytes)
            dense_splitk *= 2;
    }
    if (dense_splitk > 1) {
        t.splitkFactor = static_cast<uint16_t>(dense_splitk);
        t.kernelSerial = DGA_KERNEL_STREAMK;
        t.blockDim = static_cast<uint32_t>(blocks) * t.splitkFactor;
    } else if (!dense_splitk && groups == 1 && !contiguous && blocks * 4 <= pf.coreNum * 3 && kb >= 8) {
        uint32_t s = std::min<uint32_t>({pf.coreNum * 2 / static_cast<uint32_t>(blocks), kb / 4, 32u});
        const uint64_t operand_bytes = static_cast<uint64_t>(t.m + t.n) * t.k;
        while (s > 1 && static_cast<uint64_t>(s) * t.m * t.n * 8 * 2 > operand_bytes) --s;  // slab write + read <= half the operand read
        if (s > 1) {
            const uint32_t per = ceil_div(kb, s);
            t.splitkFactor = static_cast<uint16_t>(ceil_div(kb, per));
            t.kernelSerial = DGA_KERNEL_STREAMK;
            t.blockDim = static_cast<uint32_t>(blocks) * t.splitkFactor;
        }
    }
    if (groups == 1 && !contiguous) apply_tail_split(t, pf.coreNum);
    // raster: walk `swizzleOffset` tile-rows together so that the tiles an XCD runs AT THE SAME TIME (its CUs x
    // workgroups per CU, consecutive in the raster) are a near-square patch sharing A and B panels in its L2.  (Sizing
    // the patch by the XCD's whole share of the grid instead gave 8 / 16 where the sweep finds 4: 8192^3 472 -> 440 us.)
    const uint32_t lds_pick = static_cast<uint32_t>(pick.lds) / 2 * t.stages;
    const uint32_t wpc = std::max<uint32_t>(1, std::min<uint32_t>(pf.l1Size / std::max(1u, lds_pick), 2048 / (pick.wm * pick.wn * 64)));
    const uint32_t per_xcd = std::max<uint32_t>(1, static_cast<uint32_t>(static_cast<uint64_t>(t.blockDim) / std::max(1u, pf.xcdNum)));
    const uint32_t conc = std::min<uint32_t>(per_xcd, pf.coreNum / std::max(1u, pf.xcdNum) * wpc);
    // (a patch of gm x (conc / gm) tiles fetches gm * BM + (conc / gm) * BN operand rows into the XCD's L2: the minimum is at
    //  gm^2 = conc * BN / BM -- square in ROWS, not in tiles.  128 x 256 tiles, one per CU: gm 4 -> 8 takes 29 MB off the 170 MB
    //  configs[2] moves per launch at the same time, profiles/r05_raster_traffic.txt; 8 XCDs x (1024 + 1024) rows x K is the floor)
    uint32_t gm = 1;
    while (static_cast<uint64_t>(gm * 2) * (gm * 2) * t.m1 <= static_cast<uint64_t>(conc) * t.n1 && gm * 2 <= tiles_m) gm *= 2;
    // contiguous-grouped layout: tile rows of different groups share no B panel, so a band should not be taller than a
    // group (one 128-row block per group: walk along N, the group's tiles then share its A panel and stream its B once:
    // 933 -> 826 us at 256 groups x 128 rows; scripts/contig_stream.py)
    if (contiguous) {
        const uint32_t rows_per_group = std::max(1u, t.m / std::max(1u, t.groups) / std::max<uint32_t>(1, t.m1));
        while (gm > 1 && gm > rows_per_group) gm /= 2;
    }
    t.swizzleOffset = static_cast<uint8_t>(std::min<uint32_t>(gm, 255));
    t.ldsBytes = menu_lds_bytes(t);   // of the build that will run (stage count and wave grid are settled by now)
    prefer_loader_waves(t);
    // Decode rows (M <= 16) on a weight matrix of up to 10240 rows with 2048 <= K <= 18432: the one-launch workgroup split-K on
    // per-wave LDS-DMA rings (kernelSerial 6, stages 3 names that build; csrc/gemm_fp8_wsk_kernel.hpp).  Cold, it was the fastest
    // kernel on 64 of the 68 such shapes of the decode grids, by 3-34 % (profiles/r04_sweep_wskd/table.txt): no combine launch, no
    // slab.  Wider matrices need three or more passes per workgroup (each re-streams the A rows), shorter K leaves the eight waves
    // a k block or two each, more rows double the A share of every stage: the tile kernels keep those.
    // $DGA_NO_WSK_PICK = 1 keeps the tile kernels (A/B scripts).
    static const bool no_wsk = [] { const char *e = std::getenv("DGA_NO_WSK_PICK"); return e && std::atoi(e) != 0; }();
    // (wider matrices, up to 32768 rows with K >= 4096, take its continuous-ring build: 3-7 % ahead on 16384 / 18432 x 7168 and
    //  28672 x 4096, profiles/r04_sweep_wskd/table_wide.txt)
    // (round 4's fourth decode grid, eight more (N, K) of common models: ahead too at K = 18944 -- 148 k blocks, 3584 x 18944 0.85-0.90
    //  of the best tile plan up to 8 rows -- and on 37888 x 3584 (0.89-0.95); level on 53248 x 16384; behind once a slice is very
    //  long on a narrow matrix, 5120 x 27648 1.03-1.34: profiles/r04_sweep_decode4/)
    if (!no_wsk && groups == 1 && !contiguous && t.m <= 16 && (t.k % 16) == 0 && kb >= 16 && kb <= 160 &&
        (t.n <= 10240 || (t.n <= 65536 && kb >= 24))) {
        t.kernelSerial = DGA_KERNEL_SPLITK_WORKGROUP;
        t.m1 = 16; t.n1 = 128; t.k1 = 128;
        t.splitkFactor = 1; t.stages = 3; t.wavesM = 1; t.wavesN = 4; t.dispatchPolicyTag = DGA_POLICY_PLAIN; t.swizzleOffset = 1;
        t.blockDim = std::min<uint32_t>(ceil_div(t.n, 16), pf.coreNum);
        t.ldsBytes = menu_lds_bytes(t);   // (of the 16 x 128 tile a shape the kernel does not take falls back to: what a cache file gives back)
    }
    // Very deep K on a raster of at most 64 tiles of 256 x 256 (a quarter of the CUs): Stream-K in ONE launch (kernelSerial 7,
    // gemm_fp8_streamk_kernel.hpp) -- every tile cut into 4..16 k ranges, partials added in k order inside the launch -- instead of the
    // split-K pair with its fp32 slabs and combine launch: 256 x 4096 x 32768 69.6 -> 54.7 us, 512 x 7168 x 32768 153.9 -> 141.8,
    // 768 x 4096 x 32768 108.6 -> 105.5, 1024 x 4096 x 32768 131.1 -> 125.4; at K = 16384 / 18432 it is between -8 and +8 % and keeps
    // the pair (profiles/r05_streamk_class_sweep.txt).  The reference's rule for its Stream-K: select_kernel.cpp:303-331 (k > 3072
    // and a raster that leaves cores idle).
    if (groups == 1 && !contiguous && t.k >= 32768 && (t.k % 128) == 0 && (t.m % 256) == 0 && (t.n % 256) == 0 &&
        static_cast<uint64_t>(t.m / 256) * (t.n / 256) <= 64) {
        t.m1 = 256; t.n1 = 256; t.k1 = 128; t.wavesM = 4; t.wavesN = 2; t.stages = 2; t.splitkFactor = 1;
        t.dispatchPolicyTag = DGA_POLICY_CONTINUOUS; t.kernelSerial = DGA_KERNEL_STREAMK_ONE_LAUNCH;
        t.blockDim = pf.coreNum; t.swizzleOffset = 1;
        t.ldsBytes = menu_lds_bytes(t);
    }
}

// ---- CSV-backed (m,n,k)-keyed cache ---------------------------------------------------------
static const char *kCsvHead[] = {"m", "n", "k", "m1", "n1", "k1", "kernelSerial",
                                 "paddingTagA", "paddingTagB", "paddingTagC", "blockDim"};
constexpr int kCsvCols = 11;
// CDNA4 columns behind the reference's eleven (csv.cpp:23-26): a file that has them round-trips a tiling completely
// (split-K factor, LDS stages, raster group, wave grid, schedule); a reference-format file is still read and appended
// to in its own format.
static const char *kCsvExt[] = {"splitkFactor", "stages", "swizzleOffset", "wavesM", "wavesN", "dispatchPolicyTag"};
constexpr int kCsvExtCols = 6;
// ... and two more that key grouped problems (absent = a dense row): groups (> 1: masked grouped with m = m_max; with
// contiguous = 1: the number of B matrices of the contiguous-grouped layout)
static const char *kCsvGrp[] = {"groups", "contiguous"};
// ... and one that names the compiled build (dga_tiling_t.build, ABI 7; absent = 0).  Files from before it existed carried build names
// in magic `stages` values (1, 4..8): such a row is read as (build = stages, stages = 3).
static const char *kCsvBuild = "build";

class Cache {
public:
    static Cache &instance()
    {
        static Cache c;
        return c;
    }
    int open(const char *path)
    {
        std::lock_guard<std::mutex> lk(mu_);
        data_.clear();
        path_.clear();
        ext_ = grp_ = bld_ = false;
        if (!path || !*path) return DGA_OK;
        std::ifstream in(path);
        std::string line;
        bool have_head = false;
        std::vector<std::string> head;
        if (in.is_open() && std::getline(in, line)) {
            head = split(line);
            have_head = !head.empty();
            std::map<std::string, size_t> col;
            for (size_t i = 0; i < head.size(); ++i) col[head[i]] = i;
            for (const char *h : kCsvHead)
                if (!col.count(h)) return DGA_E_IO;
            ext_ = true;
            for (const char *h : kCsvExt) ext_ = ext_ && col.count(h);
            grp_ = ext_ && col.count(kCsvGrp[0]) && col.count(kCsvGrp[1]);
            bld_ = grp_ && col.count(kCsvBuild);
            while (std::getline(in, line)) {
                if (line.empty()) continue;
                const auto cells = split(line);
                auto get = [&](const char *name) -> uint32_t {
                    const size_t i = col[name];
                    if (i >= cells.size()) return 0;
                    char *end = nullptr;
                    const unsigned long v = std::strtoul(cells[i].c_str(), &end, 10);
                    return end == cells[i].c_str() ? 0 : static_cast<uint32_t>(v);
                };
                Entry e{};
                e.m1 = get("m1"); e.n1 = get("n1"); e.k1 = get("k1"); e.serial = get("kernelSerial");
                e.pa = get("paddingTagA"); e.pb = get("paddingTagB"); e.pc = get("paddingTagC");
                e.block_dim = get("blockDim");
                // optional CDNA4 columns (a sweep writes them; a reference-format file simply lacks them)
                auto opt = [&](const char *name) -> uint32_t { return col.count(name) ? get(name) : 0; };
                e.splitk = opt("splitkFactor"); e.stages = opt("stages"); e.raster = opt("swizzleOffset");
                e.waves_m = opt("wavesM"); e.waves_n = opt("wavesN"); e.policy = opt("dispatchPolicyTag");
                e.has_policy = col.count("dispatchPolicyTag") != 0;
                e.build = opt(kCsvBuild);
                if (e.stages == 1 || (e.stages >= 4 && e.stages <= 9)) {   // a build name of ABI <= 6
                    if (!e.build) e.build = e.stages;
                    e.stages = 3;
                }
                const uint32_t groups = std::max(1u, opt("groups")), contiguous = opt("contiguous") ? 1u : 0u;
                // rows timed under the bf16-exact policy (dispatchPolicyTag 7, with or without the power-of-two-scales flag) are a class of
                // their own: the policy has its own menu, so a shape may hold one row per class and neither shadows the other
                const uint32_t bx = (e.policy & 15u) == DGA_POLICY_BF16_EXACT ? kBxClass : 0u;
                data_[key_of(get("m"), get("n"), get("k"), groups, contiguous | bx)] = e;
            }
        }
        in.close();
        if (!have_head) {  // new or empty file: write the header row (csv.cpp InitRowHead)
            std::ofstream out(path);
            if (!out.is_open()) return DGA_E_IO;
            for (int i = 0; i < kCsvCols; ++i) out << (i ? "," : "") << kCsvHead[i];
            for (int i = 0; i < kCsvExtCols; ++i) out << "," << kCsvExt[i];
            out << "," << kCsvGrp[0] << "," << kCsvGrp[1] << "," << kCsvBuild << "\n";
            ext_ = grp_ = bld_ = true;
        }
        path_ = path;
        return DGA_OK;
    }
    void clear()
    {
        std::lock_guard<std::mutex> lk(mu_);
        data_.clear();
    }
    int size()
    {
        std::lock_guard<std::mutex> lk(mu_);
        return static_cast<int>(data_.size());
    }
    // *swept = the entry carries the CDNA4 columns of a sweep (complete as it stands); *timed_policy = its file also named the
    // dispatchPolicyTag, i.e. the schedule is the one the sweep timed
    bool get(dga_tiling_t &t, bool *swept, bool *timed_policy, bool bf16_exact_class = false)
    {
        std::lock_guard<std::mutex> lk(mu_);
        const uint32_t bx = bf16_exact_class ? kBxClass : 0u;
        auto it = data_.find(key_of(t.m, t.n, t.k, t.groups, (t.contiguous ? 1u : 0u) | bx));
        bool bucketed = false;
        // (fast class only: the bf16-exact policy's rows are exact-shape hits -- its decode rows go to the workgroup split-K by rule)
        if (it == data_.end() && !bf16_exact_class && t.groups <= 1 && !t.contiguous && t.m >= 1 && t.m <= 128) {
            // a decode batch is any M <= 128 on a handful of (N, K): what a short-M tiling depends on is the tile height that covers M
            // and the (N, K) stream, so a miss falls back to the swept row of the same (N, K) at the next row count of the decode
            // grid (harness/sweep.py --cold over M in {1, 4, 8, 16, 32, 48, 64, 96, 128}: profiles/r04_sweep_decode)
            static constexpr uint32_t kDecodeRows[] = {1, 4, 8, 16, 32, 48, 64, 96, 128};
            for (uint32_t mb : kDecodeRows) {
                if (mb < t.m) continue;
                it = data_.find(key_of(mb, t.n, t.k, 1u, bx));
                if (it != data_.end() && it->second.stages != 0) { bucketed = true; break; }
                it = data_.end();
            }
        }
        if (it == data_.end()) return false;
        const Entry &e = it->second;
        t.m1 = e.m1; t.n1 = e.n1; t.k1 = e.k1; t.kernelSerial = e.serial;
        t.paddingTagA = e.pa; t.paddingTagB = e.pb; t.paddingTagC = e.pc; t.blockDim = e.block_dim;
        t.splitkFactor = e.splitk ? static_cast<uint16_t>(e.splitk) : 1;
        t.stages = static_cast<uint8_t>(e.stages); t.wavesM = static_cast<uint8_t>(e.waves_m);
        t.wavesN = static_cast<uint8_t>(e.waves_n); t.dispatchPolicyTag = static_cast<uint8_t>(e.policy);
        t.build = static_cast<uint8_t>(e.build);
        if (e.raster) t.swizzleOffset = static_cast<uint8_t>(e.raster);
        if (bucketed && t.m1 && t.n1)   // the neighbour's grid is not this problem's
            t.blockDim = ((t.m + t.m1 - 1) / t.m1) * ((t.n + t.n1 - 1) / t.n1) * std::max<uint32_t>(1, t.splitkFactor);
        *swept = e.stages != 0;
        *timed_policy = e.has_policy;
        return true;
    }
    void put(const dga_tiling_t &t)
    {
        std::lock_guard<std::mutex> lk(mu_);
        const auto key = key_of(t.m, t.n, t.k, t.groups, (t.contiguous ? 1u : 0u) | ((t.dispatchPolicyTag & 15u) == DGA_POLICY_BF16_EXACT ? kBxClass : 0u));
        if (data_.count(key)) return;
        // the contiguous layout's row count changes from call to call in prefill serving: its tilings are keyed -- in memory
        // and in the file -- by the BUCKETED row count (key_of: a handful of rows per (n, k, groups)), and the map stops
        // growing at a bound
        if (t.contiguous && data_.size() >= kMaxEntries) return;
        Entry e{t.m1, t.n1, t.k1, t.kernelSerial, t.paddingTagA, t.paddingTagB, t.paddingTagC, t.blockDim};
        e.splitk = t.splitkFactor; e.stages = t.stages; e.raster = t.swizzleOffset; e.waves_m = t.wavesM;
        e.waves_n = t.wavesN; e.policy = t.dispatchPolicyTag; e.build = t.build;
        data_[key] = e;
        if (!path_.empty() && (grp_ || (t.groups <= 1 && !t.contiguous))) {  // a file without the group columns: dense rows only
            std::ofstream out(path_, std::ios::app);
            if (out.is_open()) {
                out << std::get<0>(key) << ',' << t.n << ',' << t.k << ',' << t.m1 << ',' << t.n1 << ',' << t.k1 << ','
                    << unsigned(t.kernelSerial) << ',' << unsigned(t.paddingTagA) << ',' << unsigned(t.paddingTagB)
                    << ',' << unsigned(t.paddingTagC) << ',' << t.blockDim;
                if (ext_)
                    out << ',' << unsigned(t.splitkFactor) << ',' << unsigned(t.stages ? t.stages : 2) << ','
                        << unsigned(t.swizzleOffset) << ',' << unsigned(t.wavesM) << ',' << unsigned(t.wavesN) << ','
                        << unsigned(t.dispatchPolicyTag);
                if (grp_) out << ',' << t.groups << ',' << unsigned(t.contiguous ? 1 : 0);
                if (bld_) out << ',' << unsigned(t.build);
                out << "\n";
            }
        }
    }

private:
    struct Entry { uint32_t m1, n1, k1, serial, pa, pb, pc, block_dim, splitk = 0, stages = 0, raster = 0, waves_m = 0, waves_n = 0, policy = 0, build = 0; bool has_policy = false; };
    static constexpr size_t kMaxEntries = 16384;
    // (m, n, k, groups, contiguous); the contiguous layout's m (total rows) is bucketed to 128 x a power of two: what the
    // tiling depends on is the rows per group against the tile heights, not the exact count
    static constexpr uint32_t kBxClass = 2u;   // bit 1 of the key's layout word: a row of the bf16-exact policy's menu
    static std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t> key_of(uint32_t m, uint32_t n, uint32_t k, uint32_t groups,
                                                                               uint32_t contiguous)
    {
        if (contiguous & 1u) {
            uint32_t blocks = (m + DGA_CONTIGUOUS_M_ALIGNMENT - 1) / DGA_CONTIGUOUS_M_ALIGNMENT, b = 1;
            while (b < blocks && b < (1u << 24)) b <<= 1;
            m = b * DGA_CONTIGUOUS_M_ALIGNMENT;
        }
        return std::make_tuple(m, n, k, std::max(1u, groups), contiguous);
    }
    Cache()
    {
        const char *p = std::getenv("DGA_CACHE_FILE_PATH");
        if (!p || !*p) p = std::getenv("CACHE_FILE_PATH");  // the reference's variable (cache.cpp:24)
        if (p && *p) {
            if (open(p) != DGA_OK) std::fprintf(stderr, "[DGA] [ERROR] Create file cache failed.\n");
            return;
        }
        // No cache file requested: preload (read-only, never appended to) the swept table shipped next to the
        // library, tuned/mi355x.csv -- the output of harness/sweep.py on the reference's shape list.
        const char *off = std::getenv("DGA_NO_TUNED_TABLE");
        if (off && *off && *off != '0') return;
        Dl_info info;
        if (dladdr(reinterpret_cast<void *>(&dga_tiling_cache_size), &info) && info.dli_fname) {
            std::string dir(info.dli_fname);
            const size_t slash = dir.rfind('/');
            dir = slash == std::string::npos ? "." : dir.substr(0, slash);
            const std::string tuned = dir + "/tuned/mi355x.csv";
            std::ifstream probe(tuned);
            if (probe.good()) {
                probe.close();
                if (open(tuned.c_str()) == DGA_OK) path_.clear();  // keep the rows, forget the path: no appends
            }
        }
    }
    static std::vector<std::string> split(const std::string &line)
    {
        std::vector<std::string> out;
        std::stringstream ss(line);
        std::string cell;
        while (std::getline(ss, cell, ',')) {
            while (!cell.empty() && (cell.back() == '\r' || cell.back() == ' ')) cell.pop_back();
            out.push_back(cell);
        }
        return out;
    }
    std::mutex mu_;
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t, uint32_t>, Entry> data_;
    std::string path_;
    bool ext_ = false;   // the open file's header has the CDNA4 columns
    bool grp_ = false;   // ... and the groups / contiguous columns
    bool bld_ = false;   // ... and the build column
};

// TilingParams ctor (tiling_params.h:45-65): strides from the layouts, swizzle defaults.
int init_params(const dga_problem_t &p, dga_tiling_t &t)
{
    std::memset(&t, 0, sizeof(t));
    t.m = p.m; t.n = p.n; t.k = p.k;
    t.layoutTagA = p.layoutTagA; t.layoutTagB = p.layoutTagB; t.layoutTagC = p.layoutTagC;
    t.strideA = (p.layoutTagA == DGA_LAYOUT_COLUMN_MAJOR) ? p.m : p.k;
    t.strideB = (p.layoutTagB == DGA_LAYOUT_COLUMN_MAJOR) ? p.k : p.n;
    t.strideC = (p.layoutTagC == DGA_LAYOUT_COLUMN_MAJOR) ? p.m : p.n;
    t.swizzleOffset = 3;
    t.swizzleDirection = (p.m > p.n) ? 0 : 1;
    t.splitkFactor = 1;
    t.groups = p.groups ? p.groups : 1;
    t.contiguous = (p.flags & DGA_PROBLEM_CONTIGUOUS_M) ? 1 : 0;
    return DGA_OK;
}

// fill the CDNA4-only fields of a tiling that came from the cache / CSV (which stores m1,n1 only)
void complete_from_menu(dga_tiling_t &t)
{
    if (!t.stages) t.stages = 2;
    // a row that asks for loader waves (dispatchPolicyTag 4) names the build that has them, whatever wave grid it carries
    if (t.dispatchPolicyTag == DGA_POLICY_LOADER_WAVES || t.dispatchPolicyTag == DGA_POLICY_PERSISTENT) {
        for (int i = 0; i < variant_count(); ++i) {
            int bm, bn, wm, wn, lds;
            variant_info(i, &bm, &bn, &wm, &wn, &lds);
            if (bm != t.m1 || bn != t.n1 || variant_stages(i) != (t.stages == 3 ? 3 : 2) || !variant_has_loader_waves(i)) continue;
            t.wavesM = static_cast<uint8_t>(wm); t.wavesN = static_cast<uint8_t>(wn);
            t.ldsBytes = static_cast<uint32_t>(lds);
            return;
        }
        t.dispatchPolicyTag = DGA_POLICY_PLAIN;   // no such build for this tile: the plain loop
    }
    // first build of that tile size with that stage count (the launcher's own preference order), else any build of it
    for (int pass = 0; pass < 2; ++pass)
        for (int i = 0; i < variant_count(); ++i) {
            int bm, bn, wm, wn, lds;
            variant_info(i, &bm, &bn, &wm, &wn, &lds);
            if (bm != t.m1 || bn != t.n1 || (pass == 0 && variant_stages(i) != (t.stages == 3 ? 3 : 2))) continue;
            if (!t.wavesM) { t.wavesM = static_cast<uint8_t>(wm); t.wavesN = static_cast<uint8_t>(wn); }
            t.ldsBytes = menu_lds_bytes(t);
            return;
        }
}

}  // namespace tiling
}  // namespace dga

using namespace dga::tiling;

extern "C" {

void dga_platform_mi355x(dga_platform_t *out)
{
    if (!out) return;
    out->coreNum = 256;               // CUs (8 XCDs x 32)
    out->ubSize = 0;
    out->l1Size = 160 * 1024;         // LDS per CU
    out->l0ASize = 64 * 1024;         // not limiting: fragments stream through VGPRs
    out->l0BSize = 64 * 1024;
    out->l0CSize = 8 * 128 * 64 * 4;  // 8 waves x 128 accumulator VGPRs x 64 lanes x 4 B = one 256x256 fp32 tile
    out->xcdNum = 8;
    out->waveSize = 64;
}

void dga_platform_ascend910b(dga_platform_t *out, uint32_t core_num)
{
    if (!out) return;
    out->coreNum = core_num ? core_num : 24;  // platform_info.h:18
    out->ubSize = 192 * 1024;
    out->l1Size = 512 * 1024;
    out->l0ASize = 64 * 1024;
    out->l0BSize = 64 * 1024;
    out->l0CSize = 128 * 1024;
    out->xcdNum = 1;
    out->waveSize = 0;
}

int dga_infer_shape(const int64_t *self_shape, int self_rank, const int64_t *mat2_shape, int mat2_rank,
                    int64_t *out_shape)
{
    if (!self_shape || !mat2_shape || !out_shape) return DGA_E_NULL;
    if (self_rank != 2 || mat2_rank != 2) return DGA_E_SHAPE;
    out_shape[0] = self_shape[0];
    out_shape[1] = mat2_shape[1];
    return DGA_OK;
}

int dga_infer_dtype(int self_dtype, int mat2_dtype, int *out_dtype)
{
    if (!out_dtype) return DGA_E_NULL;
    if (self_dtype != mat2_dtype) return DGA_E_DTYPE;
    switch (self_dtype) {
        case DGA_DT_FP16: case DGA_DT_BF16: *out_dtype = self_dtype; return DGA_OK;
        case DGA_DT_FP8_E4M3FN: *out_dtype = DGA_DT_BF16; return DGA_OK;
        default: return DGA_E_DTYPE;
    }
}

int dga_select_kernel(const dga_problem_t *problem, const dga_platform_t *platform, dga_tiling_t *out)
{
    if (!problem || !out) return DGA_E_NULL;
    if (problem->layoutTagA != DGA_LAYOUT_ROW_MAJOR || problem->layoutTagB != DGA_LAYOUT_COLUMN_MAJOR ||
        problem->layoutTagC != DGA_LAYOUT_ROW_MAJOR)
        return DGA_E_SHAPE;  // the operator is NT (catlass_dynamic_matmul_tiling.cpp:83-84)
    dga_platform_t pf;
    if (platform) {
        pf = *platform;
    } else {  // the device this call runs on: a compute-partition mode exposes fewer CUs than the whole chip
        dga_platform_mi355x(&pf);
        pf.coreNum = dga::device_cus();
        pf.xcdNum = std::max<uint32_t>(2, pf.coreNum / 32);   // (1 would select the Ascend replay below)
    }
    if (!pf.coreNum) return DGA_E_RANGE;
    init_params(*problem, *out);
    if (problem->m == 0 || problem->n == 0) { out->blockDim = 0; return DGA_OK; }
    if (pf.xcdNum <= 1) {
        if (problem->k == 0) return DGA_E_SHAPE;   // the reference's padding simulator divides by K (select_kernel.cpp:147-180)
        select_reference(*out, pf);
    } else {
        if (out->contiguous) select_mi355x(*out, pf, 1, 0, true);  // one A/out matrix; groups counts the B matrices
        else select_mi355x(*out, pf, out->groups, problem->expected_m);
        if (!out->m1) return DGA_E_TILING;
    }
    return DGA_OK;
}

int dga_tiling(const dga_problem_t *problem, dga_tiling_t *out)
{
    if (!problem || !out) return DGA_E_NULL;
    init_params(*problem, *out);
    bool swept = false, timed_policy = false;
    if (Cache::instance().get(*out, &swept, &timed_policy)) {
        // A reference-format file (or one written by hand) may name what this library's menu does not hold: the reference's own
        // fixture row `512,512,512,128,256,256,0,...` carries k1 = 256, its kernel type 3 (PaddingStreamK) has no build here.  The
        // kernels step one 128-wide scale block at a time whatever k1 says, so a cached row is normalised onto the menu -- as its
        // CDNA4-only fields are re-derived below -- instead of coming back as a tiling that dga_tiling_check refuses
        // (csv_test.cpp:33-35 rows; tests/test_tiling.py, tests/test_tiling_check.py).
        if (out->k1 != 0 && out->k1 != 128) out->k1 = 128;
        switch (out->kernelSerial) {
            case DGA_KERNEL_COMMON: case DGA_KERNEL_SMALL: case DGA_KERNEL_PADDING_COMMON: case DGA_KERNEL_STREAMK:
            case DGA_KERNEL_STREAMK_TAIL: case DGA_KERNEL_SPLITK_WORKGROUP: case DGA_KERNEL_STREAMK_ONE_LAUNCH: break;
            default: out->kernelSerial = DGA_KERNEL_COMMON;
        }
        if (out->contiguous)   // a bucketed key: the workgroup count follows this call's row count
            out->blockDim = static_cast<uint32_t>(static_cast<uint64_t>((out->m + out->m1 - 1) / std::max<uint32_t>(1, out->m1)) *
                                                  ((out->n + out->n1 - 1) / std::max<uint32_t>(1, out->n1)) *
                                                  (out->m1 > DGA_CONTIGUOUS_M_ALIGNMENT ? 2 : 1));
        if (swept) {  // a swept entry is complete: the build the sweep timed (a file from before the policy column existed, or a
                      // reference-format one, is upgraded to the loader waves where the tile has such a build)
            complete_from_menu(*out);
            dga::prefer_loader_waves(*out, !timed_policy);
            return DGA_OK;
        }
        // A reference-format CSV stores the reference's columns only; the CDNA4-only fields (waves, stages, LDS bytes,
        // raster group) are re-derived.  If the cached tile is the one the heuristic would pick, take the heuristic's
        // build of it; a tile that came from a sweep / hand-written CSV gets the menu's first build of that size.
        dga_tiling_t fresh;
        if (dga_select_kernel(problem, nullptr, &fresh) == DGA_OK && fresh.m1 == out->m1 && fresh.n1 == out->n1) {
            out->wavesM = fresh.wavesM; out->wavesN = fresh.wavesN; out->stages = fresh.stages;
            out->ldsBytes = fresh.ldsBytes; out->swizzleOffset = fresh.swizzleOffset;
        } else {
            complete_from_menu(*out);
            out->swizzleOffset = 4;
        }
        dga::prefer_loader_waves(*out);
        return DGA_OK;
    }
    // cache miss: the learned predictor where it applies (it starts from, and falls back to, the heuristic)
    int rc = dga_select_kernel_with_predictor(problem, out, nullptr, nullptr);
    if (rc != DGA_OK) return rc;
    Cache::instance().put(*out);
    return DGA_OK;
}

// Tiling of the bf16-exact policy (dispatchPolicyTag 7).  Its menu (dga_launch_menu_e.hip) and its costs are not the fast
// path's: one 8-wave build (128x256) and four 4-wave builds whose single wave per SIMD issues conversions and promotions at
// about 0.6-0.75 of the rate, and a split-K combine that costs a launch.  The fast path's tuned tile mapped onto that menu
// was 10-40 % off on mid-M shapes (512..1024 x 4096 x 7168 took the 4-wave 128x128 build; profiles/r03_bx_tile_sweep.txt), so
// dense problems get their own pick from a small cost model fitted to that sweep: rounds x (per-workgroup time) + combine.
// Grouped layouts keep the fast tiling's tile (its height is dictated by the layout).
int dga_tiling_bf16_exact(const dga_problem_t *problem, dga_tiling_t *out)
{
    if (!problem || !out) return DGA_E_NULL;
    {   // The cache first, as the reference's op does on every call (select_kernel.cpp:371-378, cache.cpp:69-100): a row the sweep timed
        // under THIS policy (dispatchPolicyTag 7 in tuned/mi355x.csv or in the caller's $DGA_CACHE_FILE_PATH; harness/sweep.py --arith
        // bf16_exact) is what a default call runs, before any cost model
        init_params(*problem, *out);
        bool swept = false, timed_policy = false;
        if (out->m && out->n && Cache::instance().get(*out, &swept, &timed_policy, /*bf16_exact_class*/ true) && out->m1 && out->n1) {
            if (out->k1 != 0 && out->k1 != 128) out->k1 = 128;
            if (!out->stages) out->stages = 3;
            if (!out->splitkFactor) out->splitkFactor = 1;
            out->wavesM = out->wavesN = 0;      // (this policy's builds are named by tile and `build`)
            if (!out->blockDim) out->blockDim = std::max<uint32_t>(1, out->groups) * ((out->m + out->m1 - 1) / out->m1) * ((out->n + out->n1 - 1) / out->n1) * out->splitkFactor;
            return DGA_OK;
        }
    }
    int rc = dga_tiling(problem, out);
    if (rc != DGA_OK) return rc;
    out->dispatchPolicyTag = DGA_POLICY_BF16_EXACT;
    // Masked grouped layout: this policy's loop multiplies every row of its tile (no per-m-tile skipping as in the fast kernels), so the
    // tile height follows the caller's hint where it says the experts are nearly empty -- rows present 0..16: 832 -> 597 us on
    // 256 x (128, 7168, 2048) with 32 x 128 tiles, 0..32: 886 -> 628, 0..64: 938 -> 734 with 64 x 256 (scripts/grouped_decode_bf16x.py,
    // profiles/r04_grouped_decode_bf16x.txt).  A hint that is too low costs time only (every further tile row of an expert streams its
    // weights again), never rows: the raster still covers m_max.
    // Masked grouped layout, experts of more than 64 rows allocated: the layout's own kernel on the 128 x 256 tile (gemm_fp8_bf16x_grouped_kernel.hpp:
    // rows that do not exist are not multiplied, at 16-row granularity, and two k blocks of the weight stream are in flight) whatever
    // the hint says -- 256 x (128, 7168, 2048): random masks 865-900 -> 761-796 us, 0..16 rows 674 (617 with the hint's 32 x 128 tiles)
    // -> 522, 0..64 rows 735 -> 600, full mask level (profiles/r06_grouped_masks.txt).
    if (std::max<uint32_t>(1, out->groups) > 1 && !out->contiguous && out->m > 64 && out->k >= 256 && (out->k % 16) == 0) {
        out->m1 = 128; out->n1 = 256;
        out->stages = 3; out->wavesM = 0; out->wavesN = 0; out->splitkFactor = 1; out->kernelSerial = DGA_KERNEL_COMMON;
        out->build = DGA_BUILD_BX_GROUPED;
        out->swizzleOffset = 1;
        out->blockDim = out->groups * ((out->m + out->m1 - 1) / out->m1) * ((out->n + out->n1 - 1) / out->n1);
        return DGA_OK;
    }
    if (std::max<uint32_t>(1, out->groups) > 1 && !out->contiguous && problem->expected_m > 0 && problem->expected_m <= 64 && out->m > 64 &&
        out->k >= 128 && (out->k % 16) == 0) {
        out->m1 = problem->expected_m <= 32 ? 32 : 64;
        out->n1 = problem->expected_m <= 32 ? 128 : 256;
        out->stages = 3; out->wavesM = 0; out->wavesN = 0; out->splitkFactor = 1; out->kernelSerial = DGA_KERNEL_COMMON;
        out->blockDim = out->groups * ((out->m + out->m1 - 1) / out->m1) * ((out->n + out->n1 - 1) / out->n1);
        return DGA_OK;
    }
    if (std::max<uint32_t>(1, out->groups) > 1 || out->contiguous || !out->m || !out->n || out->k < 128) {
        // (K % 16 != 0 takes the cost model below like any other K: the padding pass in front of the kernel is the same for every tile --
        //  8 x 7168 x 18433 75.6 -> 63.2 us on the workgroup split-K, the prefill-sized odd shapes unchanged)
        // grouped layouts keep the fast tiling's tile -- but not its wave layout: 2 x 2 waves would name this policy's 4-wave IMAGE build
        // (dga_launch.hip), 4-10 % behind the in-register build on the grouped stream and without its row skipping
        out->wavesM = out->wavesN = 0;
        return DGA_OK;
    }
    using namespace dga::tiling;
    const uint32_t cus = dga::device_cus(), kb = (out->k + 127) / 128;
    double best = 1e300;
    int bm = out->m1, bn = out->n1;
    uint32_t sk = 1;
    for (int i = 0; i < kBf16xModel.n_tiles; ++i) {
        DenseTileCost c = kBf16xModel.tiles[i];
        if (c.bm == 32 && out->m > 256) c.us_per_kblock = kBf16x32RowTallUs;
        for (uint32_t s : kDenseSplits) {
            if (s > 1 && (kb < 4 * s || static_cast<uint64_t>(s) * out->m * out->n * 4 > kMaxSlabBytes)) continue;
            uint32_t s_eff = 1;
            double us = dense_cost_us(kBf16xModel, out->m, out->n, out->k, c, s, 3, cus, 160 * 1024, &s_eff);
            us *= 1.0 + 1e-3 * (c.bm > static_cast<int>(out->m) ? (c.bm - static_cast<int>(out->m)) / 16.0 : 0.0);
            if (us < best) { best = us; bm = c.bm; bn = c.bn; sk = s_eff; }
        }
    }
    // Short-M rows on the tiles of which two workgroups share a CU (64x128, 32x128: 75 / 63 KB of LDS): the fitted model stops splitting
    // once every CU has a workgroup, but these rows are streams and the second workgroup per CU is more bytes in flight -- cold,
    // 64 x 7168 x 18432 48.2 -> 40.1 us and 64 x 7168 x 16384 43.9 -> 36.3 with 8 splits instead of 4, 128 x 4096 x 7168 23.9 -> 22.5
    // (scripts/bf16x_decode_sweep.py).  Double the split while the grid stays within two workgroups per CU and a split keeps >= 4 k blocks.
    if (out->m <= 128 && bn == 128 && bm <= 64 && kb >= 16) {   // (K < 2048: 64 x 24576 x 1536 is best unsplit)
        const uint64_t tiles0 = static_cast<uint64_t>((out->m + bm - 1) / bm) * ((out->n + bn - 1) / bn);
        while (tiles0 * sk * 2 <= 2ull * cus && kb / (2 * sk) >= 4 && static_cast<uint64_t>(2 * sk) * out->m * out->n * 4 <= kMaxSlabBytes) sk *= 2;
    }
    out->m1 = static_cast<uint16_t>(bm); out->n1 = static_cast<uint16_t>(bn);
    out->splitkFactor = static_cast<uint16_t>(sk);
    out->kernelSerial = sk > 1 ? DGA_KERNEL_STREAMK : DGA_KERNEL_COMMON;
    out->stages = 3; out->wavesM = 0; out->wavesN = 0;
    const uint64_t tiles = static_cast<uint64_t>((out->m + bm - 1) / bm) * ((out->n + bn - 1) / bn);
    out->blockDim = static_cast<uint32_t>(tiles * sk);
    // (tails per spare workgroup of the cut's second form, as gemm_fp8_bf16x_streamk_kernel.hpp bx_streamk_plan computes it)
    const uint64_t rem = tiles % cus, spare_t = (rem * 2 > cus) ? (rem + (cus - rem) - 1) / (cus - rem) : 1;
    if (bm == 128 && bn == 256 && sk == 1 && tiles >= 2 * static_cast<uint64_t>(cus) && rem > 0 && kb >= 48 && spare_t <= 4) {
        // Rasters of at least two rounds with a partial last one: Stream-K in one launch (kernelSerial 7, gemm_fp8_bf16x_streamk_kernel.hpp)
        // -- the whole rounds as the persistent kernel runs them, the last round cut along K.  3511 x 6151 x 8191 (2.73 rounds) 380 ->
        // 368 us, 1024 x 18432 x 7168 (2.25 rounds) 233 (the launch pair below) -> 227; below two rounds the cut does not pay: at 1.125
        // rounds it ties the pair, at 1.75 and at 0.78 it LOSES 2-6 % (with every CU busy on two k fronts a k block takes 2.0 us
        // instead of 1.8, and 200 adding workgroups read their partials in one burst at the end: profiles/r06_bx_streamk.txt), at
        // half a round the 128 x 128 tiles are 4 % ahead; and the adding pass at the end is a fixed cost, so K has to be deep: at K = 2048 /
        // 4096 / 5120 the cut LOSES 10 / 8 / 3 % (4096 x 7168 x 2048, 6016 x 4096 x 4096, 5120^3: profiles/r06_bx_regret.txt) -- from
        // K = 6144 on, and with at most four tails per spare workgroup.  The reference's rule: more blocks than cores with a remainder
        // below 0.8 of the cores and k > 3072 (op_host/op_tiling/select_kernel.cpp:303-331).
        out->kernelSerial = DGA_KERNEL_STREAMK_ONE_LAUNCH;
        out->blockDim = cus;
    } else
    if (bm == 128 && bn == 256 && sk == 1 && tiles > cus && tiles % cus > 0 && (tiles % cus) * 2 <= cus) {
        // the cost above counted the last partial round in quarter tiles: name that launch pair (1024 x 18432 x 7168, 2.25 rounds:
        // 276 -> 241 us; 2304 x 4096 x 7168, 1.125 rounds, 168 on 128 x 128 tiles -> 139; same bytes as the single launch)
        out->kernelSerial = DGA_KERNEL_STREAMK_TAIL;
        out->blockDim = static_cast<uint32_t>(tiles - tiles % cus + 4 * (tiles % cus));
    }
    {   // raster group: the XCD's concurrent patch square in operand ROWS (see select_mi355x); 4096^3 on 128 x 256 tiles:
        // 202 -> 169 MB of fabric traffic per launch at the same time (profiles/r05_raster_traffic.txt)
        const uint32_t tiles_m = (out->m + bm - 1) / bm;
        const uint32_t wpc = (bm * bn <= 64 * 128) ? 2 : 1;
        const uint32_t conc = static_cast<uint32_t>(std::min<uint64_t>(std::max<uint64_t>(1, tiles / 8), static_cast<uint64_t>(cus / 8) * wpc));
        uint32_t gm = 1;
        while (static_cast<uint64_t>(gm * 2) * (gm * 2) * bm <= static_cast<uint64_t>(conc) * bn && gm * 2 <= tiles_m) gm *= 2;
        out->swizzleOffset = static_cast<uint8_t>(gm);
    }
    // 17..512 rows on a matrix that gives at most one 64 x 128 tile per CU: the one-launch split-K of that tile (kernelSerial 6 with build
    // DGA_BUILD_BX_DECODE, gemm_fp8_bf16x_dsk_kernel.hpp: two k groups per workgroup, splitkFactor workgroups per tile meeting in the
    // workspace, no combine launch), S = min(8, CUs / tiles, k blocks / 4), 6 where that is 8.  Timed as decode rows are (SURVEY 8(d):
    // operand sets rotated past the Infinity Cache) against EVERY other candidate of this policy's menu, 61 shapes
    // (scripts/r06_bx_regret.py, profiles/r06_decode_cold_table.txt): 0.76-0.98 of the best other candidate inside the rule below --
    // 128 x 4096 x 7168 21.6 -> 19.0 us, 128 x 7168 x 2048 16.5 -> 13.1, 256 x 7168 x 2048 19.6 -> 14.8, 192 x 2112 x 7168 21.0 -> 16.7,
    // 64 x 16384 x 7168 36.8 -> 30.7, 64 x 4096 x 7168 17.3 -> 16.8 -- and behind it outside: fewer than 24 tiles (64 x 2112 x 7168 +3 %:
    // the ~4 us hand-over between workgroups is not bought back), K below 1536, short K on few tiles (64 x 4096 x 2048 +4 %), more than
    // 30 k blocks per k group (128 x 7168 x 18432 +4 %; 16 above 128 rows: 256 x 5120 x 5120 +7 %).  Warm (the weights in the Infinity
    // Cache, profiles/r06_decode_grid.txt) the 49..64-row picks are level to 10 % behind the tile kernels; decode weights are not warm.
    // 257..512 rows (timed warm, nine shapes): 0.86-1.01 of the best other candidate (320 x 2112 x 7168 22.0 -> 18.8 us, 512 x 2048 x 7168 26.4 -> 24.0).
    static const bool no_dsk = [] { const char *e = std::getenv("DGA_NO_DSK_PICK"); return e && std::atoi(e) != 0; }();
    // (17..32 rows: where the matrix is too tall for the per-wave split-K below -- 32 x 24576 x 1536 15.9 -> 13.0 us, 24 x 12288 x 5120
    //  21.8 -> 20.1; on matrices of at most 8192 rows that kernel stays 7-30 % ahead)
    //  (... and where K is long: 24 x 4096 x 18432 26.6 -> 23.4, 32 x 7168 x 18432 33.2 -> 31.5)
    if (!no_dsk && (out->k % 16) == 0 && kb >= 12 && out->m <= 512 && (out->m >= 33 || (out->m >= 17 && (out->n > 8192 || kb >= 96)))) {
        const uint64_t dt = static_cast<uint64_t>((out->m + 63) / 64) * ((out->n + 127) / 128);
        if (dt >= 24 && dt <= cus && !(kb < 32 && dt < 48)) {
            const uint32_t smax = static_cast<uint32_t>(std::min<uint64_t>(std::min<uint64_t>(8, cus / dt), kb / 4));
            // (a grid on fewer than 3/4 of the CUs with long k groups: 64 x 18432 x 7168, 144 workgroups of 28 k blocks per group, is 5 %
            //  behind the tile kernel's 432 workgroups, 48 x 18432 x 7168 10 %)
            const bool thin = dt * smax * 4 < 3ull * cus && kb > 32u * smax;
            if (smax >= 1 && !thin && kb <= (out->m > 128 ? 32u : 60u) * smax) {
                // (eight workgroups per tile on every CU: six measured 2-4 % ahead -- 64 x 4096 x 7168 17.1 / 17.6 us, 64 x 4096 x 4096
                //  14.0 / 14.7; on a grid that leaves CUs free eight stay ahead -- 48 x 3072 x 18432 23.1 / 25.4)
                const uint32_t s = (smax == 8 && dt * 8 > 7ull * cus / 8) ? 6 : smax;
                out->kernelSerial = DGA_KERNEL_SPLITK_WORKGROUP;
                out->build = DGA_BUILD_BX_DECODE;
                out->m1 = 64; out->n1 = 128;
                out->splitkFactor = static_cast<uint16_t>(s); out->stages = 3; out->swizzleOffset = 1;
                out->blockDim = static_cast<uint32_t>(dt * s);
                return DGA_OK;
            }
        }
    }
    // Decode rows: the workgroup split-K on LDS-DMA rings runs this policy's arithmetic too (gemm_fp8_wskd_kernel<..., MATH = 1>), and
    // since the stream pays for neither the conversions nor the bf16 matrix rate it takes the fast policy's time: cold, 20-44 % ahead
    // of this policy's tile kernels on 103 of 120 decode shapes (profiles/r04_wskd_cold_bf16x.txt) -- up to 16 rows wherever a wave
    // gets at least one k block and the matrix has at most 65536 rows, up to 32 rows on matrices of at most 8192.
    static const bool no_wsk = [] { const char *e = std::getenv("DGA_NO_WSK_PICK"); return e && std::atoi(e) != 0; }();
    if (!no_wsk && ((out->m <= 16 && kb >= 8 && out->n <= 65536) || (out->m <= 32 && kb >= 12 && out->n <= 8192))) {
        out->kernelSerial = DGA_KERNEL_SPLITK_WORKGROUP;
        out->m1 = out->m <= 16 ? 16 : 32; out->n1 = 128;
        out->splitkFactor = 1; out->stages = 3; out->swizzleOffset = 1;
        out->blockDim = std::min<uint32_t>((out->n + 15) / 16, cus);
    }
    return DGA_OK;
}

int dga_tiling_cache_open(const char *csv_path) { return Cache::instance().open(csv_path); }
int dga_tiling_cache_clear(void) { Cache::instance().clear(); return DGA_OK; }
int dga_tiling_cache_size(void) { return Cache::instance().size(); }

size_t dga_workspace_bytes(const dga_tiling_t *tiling)
{
    if (!tiling) return 0;
    size_t bytes = 0;
    auto add = [&](size_t b) { bytes = ((bytes + 255) & ~size_t(255)) + b; };
    const size_t groups = tiling->groups ? tiling->groups : 1;
    if (tiling->k % 16 != 0 && tiling->k > 0) {  // padded copies of A and B (rows zero-filled to a multiple of 128)
        const size_t kp = (static_cast<size_t>(tiling->k) + 127) / 128 * 128;
        add((tiling->contiguous ? 1 : groups) * tiling->m * kp);
        add(groups * tiling->n * kp);
    }
    if (tiling->splitkFactor > 1) add(static_cast<size_t>(tiling->splitkFactor) * tiling->m * tiling->n * 4);
    // Stream-K proper: one fp32 partial tile (256 x 256) per CU + the flags
    // (one slot per CU: 256 x 256 floats on the fast path, 128 x 256 under the bf16-exact policy; + the flags)
    if (tiling->kernelSerial == DGA_KERNEL_STREAMK_ONE_LAUNCH) add(static_cast<size_t>(dga::device_cus()) * (256 * 256 * 4 + 8) + 256);
    // the decode build of the workgroup split-K: one fp32 partial tile (64 x 128) per tile and split but the first + the flags
    if (tiling->build == DGA_BUILD_BX_DECODE && tiling->splitkFactor > 1) {
        const size_t tiles = (static_cast<size_t>(tiling->m) + 63) / 64 * ((static_cast<size_t>(tiling->n) + 127) / 128);
        add(tiles * (tiling->splitkFactor - 1) * (64 * 128 * 4 + 8) + 256);
    }
    return bytes ? bytes + 256 : 0;
}

const char *dga_status_string(int status)
{
    switch (status) {
        case DGA_OK: return "ok";
        case DGA_E_NULL: return "null pointer";
        case DGA_E_SHAPE: return "shape / rank / layout mismatch";
        case DGA_E_DTYPE: return "dtype mismatch";
        case DGA_E_ALIGN: return "alignment";
        case DGA_E_HIP: return "HIP runtime error";
        case DGA_E_TILING: return "no compiled kernel for this tiling";
        case DGA_E_WORKSPACE: return "workspace too small";
        case DGA_E_IO: return "file error";
        case DGA_E_RANGE: return "value out of range";
        default: return "unknown";
    }
}

int dga_abi_version(void) { return DGA_ABI_VERSION; }

}  // extern "C"
