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
//   * platform.xcdNum <= 1 (Ascend numbers): the reference's own NT tile search replayed,
//     16-element granularity, L1/L0C limits -- exists so tests can pin the restatement
//     against the reference's golden tuples (tests/golden/op_tiling_vectors.json).
//   * platform.xcdNum  > 1 (MI355X): same search skeleton (start from the aspect-ratio tile,
//     balance the block count against the core count, bound by on-chip space), but the result
//     is drawn from the compiled kernel menu (dga_launch.hip kVariants), space is LDS + VGPR
//     accumulators, K step is one 128-wide scale block, and padding variants do not exist
//     (NZ re-layout is an Ascend artefact; the LDS image is swizzled by the DMA source address).
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

// ---- handler chain, reference mode (select_kernel.cpp:270-331) ---------------------------
// The padding cost model (GetPaddingTag :82-268) prices Ascend ND->NZ re-layout on vector cores;
// it has no CDNA4 meaning.  In reference mode we evaluate only its *outcome class* that the NT
// configs of SURVEY.md 8(a7) exercise -- (NONE,NONE,NONE) -- via the cheap structural exits of that
// model (inner axis >= 32 and 16-aligned, inner axis not a >8192 multiple of 8192, n % 128 == 0 or
// small output); anything else is reported as kernelSerial 2 without attempting the cost fit.
bool padding_free_nt(const dga_tiling_t &t)
{
    const uint64_t inner = t.k;  // NT: both operands are K-contiguous
    if (inner < 8 || (inner < 32 && inner % 16 != 0)) return false;
    if (inner > 8192 && inner % 8192 == 0 && (t.m >= 2048 || t.n >= 2048)) return false;
    if (static_cast<uint64_t>(t.m) * t.n > 2048ull * 2048ull && t.n > 256 && t.n % 128 != 0) return false;
    return true;
}

void select_reference(dga_tiling_t &t, const dga_platform_t &pf)
{
    do_tiling_nt_reference(t, pf, 2);
    const uint32_t blocks = ceil_div(t.m, t.m1) * ceil_div(t.n, t.n1);
    const bool nopad = padding_free_nt(t);
    t.paddingTagA = t.paddingTagB = t.paddingTagC = DGA_PADDING_NONE;
    // Small (:278-293)
    if (nopad && blocks <= pf.coreNum && t.k <= t.k1) {
        t.kernelSerial = DGA_KERNEL_SMALL;
        t.blockDim = blocks;
        return;
    }
    // Stream-K (:303-331): best-bandwidth tile, all cores, when the tail round is < 80 % full and K is long
    {
        const uint32_t sb = ceil_div(t.m, 128) * ceil_div(t.n, 256);
        const uint32_t rem = sb % pf.coreNum;
        if (sb > pf.coreNum && sb < 8 * pf.coreNum && rem > 0 && rem < 0.8 * pf.coreNum && t.k > 3072) {
            t.m1 = 128; t.n1 = 256; t.k1 = 256;
            t.blockDim = pf.coreNum;
            t.kernelSerial = DGA_KERNEL_STREAMK;
            return;
        }
    }
    if (!nopad) {
        t.kernelSerial = DGA_KERNEL_PADDING_COMMON;
        t.blockDim = std::min(blocks, pf.coreNum);
        return;
    }
    t.kernelSerial = DGA_KERNEL_COMMON;  // :270-276
    t.blockDim = std::min(blocks, pf.coreNum);
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

void select_mi355x(dga_tiling_t &t, const dga_platform_t &pf, uint32_t groups, uint32_t expected_m)
{
    const uint32_t m_eff = (groups > 1 && expected_m) ? std::min(expected_m, t.m) : t.m;
    const uint32_t kb = ceil_div(std::max(t.k, 1u), 128);
    double best = 1e300;
    MenuEntry pick{};
    bool found = false;
    for (const MenuEntry &e : menu()) {
        if (static_cast<uint64_t>(e.lds) > pf.l1Size) continue;
        if (4ull * e.bm * e.bn > pf.l0CSize) continue;  // accumulators: JudgeSpace's L0C test on VGPRs
        const uint32_t wg_per_cu = std::max<uint64_t>(1, std::min<uint64_t>(pf.l1Size / e.lds, 2048 / (e.wm * e.wn * 64)));
        const uint64_t tiles = static_cast<uint64_t>(groups) * ceil_div(m_eff, e.bm) * ceil_div(t.n, e.bn);
        const uint64_t slots = static_cast<uint64_t>(pf.coreNum) * wg_per_cu;
        const double rounds = std::ceil(static_cast<double>(tiles) / slots);
        // co-resident workgroups share the CU's MFMA pipes
        const double share = std::min<double>(wg_per_cu, std::ceil(static_cast<double>(tiles) / pf.coreNum));
        double cost = rounds * share * kb * tile_cycles_per_kblock(e);
        // wasted rows of a partially filled tile are paid in full
        cost *= 1.0 + 1e-3 * (e.bm > m_eff ? (e.bm - m_eff) / 16.0 : 0.0);
        if (cost < best) { best = cost; pick = e; found = true; }
    }
    if (!found) { t.m1 = t.n1 = 0; return; }
    t.m1 = static_cast<uint16_t>(pick.bm); t.n1 = static_cast<uint16_t>(pick.bn); t.k1 = 128;
    t.wavesM = static_cast<uint8_t>(pick.wm); t.wavesN = static_cast<uint8_t>(pick.wn);
    t.stages = 2;
    t.ldsBytes = pick.lds;
    const uint32_t tiles_m = ceil_div(t.m, t.m1), tiles_n = ceil_div(t.n, t.n1);
    const uint64_t blocks = static_cast<uint64_t>(groups) * tiles_m * tiles_n;
    t.blockDim = static_cast<uint32_t>(blocks);
    t.paddingTagA = t.paddingTagB = t.paddingTagC = DGA_PADDING_NONE;
    // variant menu in the reference's order: Small (one tile per core, single K step) -> Common
    t.kernelSerial = (blocks <= pf.coreNum && t.k <= t.k1) ? DGA_KERNEL_SMALL : DGA_KERNEL_COMMON;
    // raster: walk `swizzleOffset` tile-rows together so that an XCD's slice of the grid
    // (blocks/8 consecutive tiles) is a near-square patch sharing A and B panels in its L2.
    const uint32_t per_xcd = std::max<uint32_t>(1, static_cast<uint32_t>(blocks / std::max(1u, pf.xcdNum)));
    uint32_t gm = 1;
    while ((gm * 2) * (gm * 2) <= per_xcd * 2 && gm * 2 <= tiles_m) gm *= 2;
    t.swizzleOffset = static_cast<uint8_t>(std::min<uint32_t>(gm, 255));
}

// ---- CSV-backed (m,n,k)-keyed cache ---------------------------------------------------------
static const char *kCsvHead[] = {"m", "n", "k", "m1", "n1", "k1", "kernelSerial",
                                 "paddingTagA", "paddingTagB", "paddingTagC", "blockDim"};
constexpr int kCsvCols = 11;

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
                data_[std::make_tuple(get("m"), get("n"), get("k"), 1u)] = e;
            }
        }
        in.close();
        if (!have_head) {  // new or empty file: write the header row (csv.cpp InitRowHead)
            std::ofstream out(path);
            if (!out.is_open()) return DGA_E_IO;
            for (int i = 0; i < kCsvCols; ++i) out << (i ? "," : "") << kCsvHead[i];
            out << "\n";
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
    bool get(dga_tiling_t &t)
    {
        std::lock_guard<std::mutex> lk(mu_);
        auto it = data_.find(std::make_tuple(t.m, t.n, t.k, t.groups));
        if (it == data_.end()) return false;
        const Entry &e = it->second;
        t.m1 = e.m1; t.n1 = e.n1; t.k1 = e.k1; t.kernelSerial = e.serial;
        t.paddingTagA = e.pa; t.paddingTagB = e.pb; t.paddingTagC = e.pc; t.blockDim = e.block_dim;
        return true;
    }
    void put(const dga_tiling_t &t)
    {
        std::lock_guard<std::mutex> lk(mu_);
        const auto key = std::make_tuple(t.m, t.n, t.k, t.groups);
        if (data_.count(key)) return;
        Entry e{t.m1, t.n1, t.k1, t.kernelSerial, t.paddingTagA, t.paddingTagB, t.paddingTagC, t.blockDim};
        data_[key] = e;
        if (!path_.empty() && t.groups <= 1) {  // the CSV schema has no group column: dense rows only
            std::ofstream out(path_, std::ios::app);
            if (out.is_open())
                out << t.m << ',' << t.n << ',' << t.k << ',' << t.m1 << ',' << t.n1 << ',' << t.k1 << ','
                    << unsigned(t.kernelSerial) << ',' << unsigned(t.paddingTagA) << ',' << unsigned(t.paddingTagB)
                    << ',' << unsigned(t.paddingTagC) << ',' << t.blockDim << "\n";
        }
    }

private:
    struct Entry { uint32_t m1, n1, k1, serial, pa, pb, pc, block_dim; };
    Cache()
    {
        const char *p = std::getenv("DGA_CACHE_FILE_PATH");
        if (!p || !*p) p = std::getenv("CACHE_FILE_PATH");  // the reference's variable (cache.cpp:24)
        if (p && *p && open(p) != DGA_OK) std::fprintf(stderr, "[DGA] [ERROR] Create file cache failed.\n");
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
    std::map<std::tuple<uint32_t, uint32_t, uint32_t, uint32_t>, Entry> data_;
    std::string path_;
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
    return DGA_OK;
}

// fill the CDNA4-only fields of a tiling that came from the cache / CSV (which stores m1,n1 only)
void complete_from_menu(dga_tiling_t &t)
{
    for (const MenuEntry &e : menu())
        if (e.bm == t.m1 && e.bn == t.n1) {
            t.wavesM = e.wm; t.wavesN = e.wn; t.stages = 2; t.ldsBytes = e.lds;
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
    if (platform) pf = *platform; else dga_platform_mi355x(&pf);
    if (!pf.coreNum) return DGA_E_RANGE;
    init_params(*problem, *out);
    if (problem->m == 0 || problem->n == 0) { out->blockDim = 0; return DGA_OK; }
    if (pf.xcdNum <= 1) {
        select_reference(*out, pf);
    } else {
        select_mi355x(*out, pf, out->groups, problem->expected_m);
        if (!out->m1) return DGA_E_TILING;
    }
    return DGA_OK;
}

int dga_tiling(const dga_problem_t *problem, dga_tiling_t *out)
{
    if (!problem || !out) return DGA_E_NULL;
    init_params(*problem, *out);
    if (Cache::instance().get(*out)) {
        complete_from_menu(*out);
        if (!out->swizzleOffset || out->groups >= 1) {
            // raster group is derived, not cached
            dga_tiling_t fresh;
            if (dga_select_kernel(problem, nullptr, &fresh) == DGA_OK && fresh.m1 == out->m1 && fresh.n1 == out->n1)
                out->swizzleOffset = fresh.swizzleOffset;
            else
                out->swizzleOffset = 4;
        }
        return DGA_OK;
    }
    int rc = dga_select_kernel(problem, nullptr, out);
    if (rc != DGA_OK) return rc;
    Cache::instance().put(*out);
    return DGA_OK;
}

int dga_tiling_cache_open(const char *csv_path) { return Cache::instance().open(csv_path); }
int dga_tiling_cache_clear(void) { Cache::instance().clear(); return DGA_OK; }
int dga_tiling_cache_size(void) { return Cache::instance().size(); }

size_t dga_workspace_bytes(const dga_tiling_t *tiling)
{
    if (!tiling) return 0;
    if (tiling->kernelSerial == DGA_KERNEL_STREAMK && tiling->splitkFactor > 1)
        return static_cast<size_t>(tiling->splitkFactor) * tiling->m * tiling->n * 4 + 4096;
    return 0;
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
