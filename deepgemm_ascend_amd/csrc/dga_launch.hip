// Launch glue of the C ABI: argument checks, tiling lookup, kernel-variant dispatch.
// Host counterpart of mmad_rtc (/root/reference/deep_gemm_ascend/framework/csrc/jit_kernels/impls/gemm.hpp:68-111):
// shapes -> tiling -> launch on the caller's stream.  What the reference does with a cmake-subprocess
// JIT and a blocking aclrtSynchronizeStream (gemm.hpp:103-110) is here an AOT template menu and an
// asynchronous launch; ACL errors that the reference prints and ignores (utils/exception.hpp:35-43)
// are returned as DGA_E_HIP.
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "dga_hip.h"
#include "dga_internal.hpp"
#include "dga_fp8_menu.hpp"
#include "gemm_fp8_aux_kernels.hpp"
#include "gemm_fp8_strict_kernel.hpp"

namespace dga {

static std::atomic<int> g_last_hip_error{0};
int record_hip(hipError_t e)
{
    if (e != hipSuccess) {
        g_last_hip_error.store(static_cast<int>(e));
        return DGA_E_HIP;
    }
    return DGA_OK;
}
uint32_t device_cus()
{
    static std::atomic<uint32_t> cache[64];
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (const uint32_t v = cache[dev].load()) return v;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    cache[dev].store(static_cast<uint32_t>(n));
    return static_cast<uint32_t>(n);
}
#define DGA_HIP_TRY(expr)                     \
    do {                                      \
        int _rc = dga::record_hip((expr));    \
        if (_rc != DGA_OK) return _rc;        \
    } while (0)

struct Variant {
    int bm, bn, wm, wn;
    int (*launch)(const GemmParams &, hipStream_t);
    int lds;
    int (*launch_pp)(const GemmParams &, hipStream_t);  // ping-pong schedule (dispatchPolicyTag 1), or null
    int (*launch_cont)(const GemmParams &, hipStream_t);  // continuous pipeline (dispatchPolicyTag 2), or null
    int stages = 2;
    int (*launch_lc)(const GemmParams &, hipStream_t) = nullptr;  // loader waves + plain loop (dispatchPolicyTag 4), or null
    int (*launch_ps)(const GemmParams &, hipStream_t) = nullptr;  // persistent loader waves (dispatchPolicyTag 5), or null
};

#define DGA_VARIANT(BM, BN, WM, WN) \
    Variant { BM, BN, WM, WN, &launch_cfg<GemmCfg<BM, BN, WM, WN>, 0>, GemmCfg<BM, BN, WM, WN>::LDS_BYTES, nullptr, nullptr }
#define DGA_VARIANT_C(BM, BN, WM, WN) \
    Variant { BM, BN, WM, WN, &launch_cfg<GemmCfg<BM, BN, WM, WN>, 0>, GemmCfg<BM, BN, WM, WN>::LDS_BYTES, nullptr, \
              &launch_cfg<GemmCfg<BM, BN, WM, WN>, 2> }
#define DGA_VARIANT_PP(BM, BN, WM, WN)                                                                       \
    Variant { BM, BN, WM, WN, &launch_cfg<GemmCfg<BM, BN, WM, WN>, 0>, GemmCfg<BM, BN, WM, WN>::LDS_BYTES, \
              &launch_cfg<GemmCfg<BM, BN, WM, WN>, 1>, &launch_cfg<GemmCfg<BM, BN, WM, WN>, 2> }

static const Variant kVariants[] = {
    DGA_VARIANT_PP(256, 256, 4, 2), DGA_VARIANT_C(128, 256, 2, 2), DGA_VARIANT_C(256, 128, 4, 1),
    DGA_VARIANT_C(128, 128, 2, 2), DGA_VARIANT_C(64, 256, 1, 4),  DGA_VARIANT(64, 128, 1, 4),
    DGA_VARIANT_C(128, 256, 2, 4),  // 8 waves on a 128x256 tile: two waves per SIMD cover each other's DMA stalls
    // three LDS stages (two refills in flight) for the HBM-bound grouped / small-M shapes
    // (8 waves first: a tiling that names no wave layout -- a swept CSV row, the predictor -- gets this build; the selector
    //  names 2x2 explicitly for the masked grouped stream)
    Variant{128, 256, 2, 4, &launch_cfg<GemmCfg<128, 256, 2, 4, 3>, 0>, GemmCfg<128, 256, 2, 4, 3>::LDS_BYTES, nullptr, nullptr, 3},
    Variant{128, 256, 2, 2, &launch_cfg<GemmCfg<128, 256, 2, 2, 3>, 0>, GemmCfg<128, 256, 2, 2, 3>::LDS_BYTES, nullptr, nullptr, 3,
            &launch_cfg<GemmCfg<128, 256, 2, 2, 3, 4>, 0>, &launch_persistent<GemmCfg<128, 256, 2, 2, 3, 4>>},
    Variant{128, 128, 2, 2, &launch_cfg<GemmCfg<128, 128, 2, 2, 3>, 0>, GemmCfg<128, 128, 2, 2, 3>::LDS_BYTES, nullptr, nullptr, 3,
            &launch_cfg<GemmCfg<128, 128, 2, 2, 3, 4>, 0>, &launch_persistent<GemmCfg<128, 128, 2, 2, 3, 4>>},
    Variant{64, 256, 1, 4, &launch_cfg<GemmCfg<64, 256, 1, 4, 3>, 0>, GemmCfg<64, 256, 1, 4, 3>::LDS_BYTES, nullptr, nullptr, 3,
            &launch_cfg<GemmCfg<64, 256, 1, 4, 3, 4>, 0>, &launch_persistent<GemmCfg<64, 256, 1, 4, 3, 4>>},

    DGA_VARIANT(32, 256, 1, 4),  DGA_VARIANT(32, 128, 1, 4),  DGA_VARIANT(16, 256, 1, 4),
    DGA_VARIANT(16, 128, 1, 4),
    // three stages for the short tiles too: decode shapes stream their weights cold, and a second refill in flight per
    // workgroup is what the HBM round trip needs (profiles/r02_steady_table.json: warm 5.5 TB/s, cold 3.8 with two stages).
    // (Five stages -- one workgroup per CU, four refills in flight -- were tried in round 4 after the 16-bit tiles gained 15-25 %
    //  from going 2 -> 4: here they change nothing or lose 3-10 % on eight cold decode shapes at 48-128 rows; two workgroups of three
    //  stages already keep enough in flight.  Not in the menu.)
    Variant{64, 128, 1, 4, &launch_cfg<GemmCfg<64, 128, 1, 4, 3>, 0>, GemmCfg<64, 128, 1, 4, 3>::LDS_BYTES, nullptr, nullptr, 3,
            &launch_cfg<GemmCfg<64, 128, 1, 4, 3, 4>, 0>, &launch_persistent<GemmCfg<64, 128, 1, 4, 3, 4>>},
    Variant{32, 256, 1, 4, &launch_cfg<GemmCfg<32, 256, 1, 4, 3>, 0>, GemmCfg<32, 256, 1, 4, 3>::LDS_BYTES, nullptr, nullptr, 3},
    Variant{32, 128, 1, 4, &launch_cfg<GemmCfg<32, 128, 1, 4, 3>, 0>, GemmCfg<32, 128, 1, 4, 3>::LDS_BYTES, nullptr, nullptr, 3},
    Variant{16, 256, 1, 4, &launch_cfg<GemmCfg<16, 256, 1, 4, 3>, 0>, GemmCfg<16, 256, 1, 4, 3>::LDS_BYTES, nullptr, nullptr, 3},
    Variant{16, 128, 1, 4, &launch_cfg<GemmCfg<16, 128, 1, 4, 3>, 0>, GemmCfg<16, 128, 1, 4, 3>::LDS_BYTES, nullptr, nullptr, 3,
            &launch_cfg<GemmCfg<16, 128, 1, 4, 3, 4>, 0>, &launch_persistent<GemmCfg<16, 128, 1, 4, 3, 4>>},
};
static constexpr int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// the bf16-exact policy's own menu (dispatchPolicyTag 7; dga_launch_menu_e.hip), tallest and widest first
struct Bf16xVariant {
    int bm, bn;
    int (*launch)(const GemmParams &, hipStream_t);
};
static const Bf16xVariant kBf16xVariants[] = {
    {128, 256, &launch_bf16x<GemmCfg<128, 256, 2, 4, 3>>}, {128, 128, &launch_bf16x<GemmCfg<128, 128, 2, 2, 3>>},
    {64, 256, &launch_bf16x<GemmCfg<64, 256, 1, 4, 3>>},   {64, 128, &launch_bf16x<GemmCfg<64, 128, 1, 4, 3>>},
    {32, 128, &launch_bf16x<GemmCfg<32, 128, 1, 4, 3>>},
};
// the tiling's (m1, n1) mapped onto that menu: the tile height rounded into {32, 64, 128}, the width kept where the height
// has a build of that width
static const Bf16xVariant *find_bf16x_variant(int m1, int n1)
{
    const int bm = m1 >= 128 ? 128 : (m1 >= 64 ? 64 : 32);
    const int bn = n1 >= 256 ? 256 : 128;
    for (const auto &v : kBf16xVariants)
        if (v.bm == bm && v.bn == bn) return &v;
    for (const auto &v : kBf16xVariants)
        if (v.bm == bm) return &v;
    return nullptr;
}

// $DGA_DEFAULT_POLICY, parsed once: 0 fast, 1 bf16_exact, 2 strict, 3 fast_ue8m0, 4 bf16_exact_ue8m0, 5 auto; -1 = a value that is none
// of these (every front end refuses it: dga_default_policy)
static const char *const kPolicyNames[] = {"fast", "bf16_exact", "strict", "fast_ue8m0", "bf16_exact_ue8m0", "auto"};
int default_policy()
{
    static const int v = [] {
        const char *e = std::getenv("DGA_DEFAULT_POLICY");
        if (!e || !*e) return 1;
        for (int i = 0; i < 6; ++i)
            if (!std::strcmp(e, kPolicyNames[i])) return i;
        return -1;
    }();
    return v;
}

int variant_count() { return kNumVariants; }
int variant_stages(int i) { return kVariants[i].stages; }
bool variant_has_loader_waves(int i) { return kVariants[i].launch_lc != nullptr; }
void variant_info(int i, int *bm, int *bn, int *wm, int *wn, int *lds)
{
    *bm = kVariants[i].bm; *bn = kVariants[i].bn; *wm = kVariants[i].wm; *wn = kVariants[i].wn;
    *lds = kVariants[i].lds;
}

// (wm, wn) == (0, 0): first variant of that tile size (the menu order is the preference order)
static const Variant *find_variant(int bm, int bn, int wm, int wn, int stages)
{
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].bm == bm && kVariants[i].bn == bn && (!wm || (kVariants[i].wm == wm && kVariants[i].wn == wn)) &&
            kVariants[i].stages == (stages == 3 ? 3 : 2))
            return &kVariants[i];
    if (stages == 3) return find_variant(bm, bn, wm, wn, 2);
    return wm ? find_variant(bm, bn, 0, 0, 2) : nullptr;
}

// Every (kernelSerial, dispatchPolicyTag, tile, wave grid, stage count, split factor) the compiled menu holds -- anything else is
// refused BEFORE any launch (dga_tiling_check; run_fp8 calls it first).  A caller's dga_tiling_t is data from outside: the
// counterpart of CatlassDynamicMatmulTilingFunc returning GRAPH_FAILED on what it cannot tile
// (/root/reference/aclnn_catlass_dynamic_matmul/op_host/catlass_dynamic_matmul_tiling.cpp:86-100).
//   tiling.build names a compiled build beside the tile and the schedule (include/dga_hip.h DGA_BUILD_*; ABI <= 6 carried these
//   names in magic values of tiling.stages); tiling.stages is the LDS stage count and nothing else: 0 (the tile's default), 2 or 3
static int check_tiling(const dga_tiling_t &t)
{
    if (t.dispatchPolicyTag & ~(DGA_POLICY_UE8M0_SCALES | 7)) return DGA_E_TILING;      // bits nobody defined
    const int tag = t.dispatchPolicyTag & 7;
    switch (t.kernelSerial) {
        case DGA_KERNEL_COMMON: case DGA_KERNEL_SMALL: case DGA_KERNEL_PADDING_COMMON: case DGA_KERNEL_STREAMK:
        case DGA_KERNEL_STREAMK_TAIL: case DGA_KERNEL_SPLITK_WORKGROUP: case DGA_KERNEL_STREAMK_ONE_LAUNCH: break;
        default: return DGA_E_TILING;
    }
    if (t.k1 != 0 && t.k1 != 128) return DGA_E_TILING;                                  // one scale block per k step
    if (t.splitkFactor > 1024) return DGA_E_RANGE;
    if (t.reserved0) return DGA_E_TILING;
    if (tag == DGA_POLICY_STRICT) return DGA_OK;                                        // takes every shape as it is
    if (t.m1 == 0 || t.n1 == 0) return DGA_E_TILING;
    if (!(t.stages == 0 || t.stages == 2 || t.stages == 3)) return DGA_E_TILING;          // LDS stages some build has
    const bool wsk = t.kernelSerial == DGA_KERNEL_SPLITK_WORKGROUP;
    if (t.build == DGA_BUILD_WSK_REGISTER && !wsk) return DGA_E_TILING;
    if (tag == DGA_POLICY_BF16_EXACT) {
        if (!find_bf16x_variant(t.m1, t.n1)) return DGA_E_TILING;
        switch (t.build) {
            case DGA_BUILD_DEFAULT: case DGA_BUILD_WSK_REGISTER: case DGA_BUILD_BX_AIMAGE: case DGA_BUILD_BX_IMAGE8: case DGA_BUILD_BX_IMAGE4:
            case DGA_BUILD_BX_PERSISTENT: case DGA_BUILD_BX_ONE_TILE: case DGA_BUILD_BX_GROUPED: break;
            case DGA_BUILD_BX_DECODE:       // the one-launch split-K of the 64 x 128 tile: a name of kernelSerial 6
                if (!wsk || t.m1 != 64 || t.n1 != 128) return DGA_E_TILING;
                break;
            default: return DGA_E_TILING;
        }
        // the quarter-tile tail and the one-launch Stream-K are builds of the 128 x 256 tile
        if ((t.kernelSerial == DGA_KERNEL_STREAMK_TAIL || t.kernelSerial == DGA_KERNEL_STREAMK_ONE_LAUNCH) && !(t.m1 >= 128 && t.n1 >= 256)) return DGA_E_TILING;
        return DGA_OK;
    }
    bool tile = false, grid = false;
    for (int i = 0; i < kNumVariants; ++i)
        if (kVariants[i].bm == t.m1 && kVariants[i].bn == t.n1) {
            tile = true;
            if (kVariants[i].wm == t.wavesM && kVariants[i].wn == t.wavesN) grid = true;
        }
    if (!tile) return DGA_E_TILING;
    // (the 256 x 256 tile on 2 x 2 waves exists as a hardware-scale build only: dga_launch_menu_j.hip)
    if (t.m1 == 256 && t.n1 == 256 && t.wavesM == 2 && t.wavesN == 2 && (t.dispatchPolicyTag & DGA_POLICY_UE8M0_SCALES) &&
        (tag == DGA_POLICY_CONTINUOUS || tag == DGA_POLICY_CONTINUOUS_PERSISTENT))
        grid = true;
    if ((t.wavesM || t.wavesN) && !grid) return DGA_E_TILING;                           // a wave grid no build of this tile has
    if (t.build != DGA_BUILD_DEFAULT && t.build != DGA_BUILD_WSK_REGISTER) return DGA_E_TILING;   // the bf16-exact policy's names
    if (tag == DGA_POLICY_PINGPONG && !(t.m1 == 256 && t.n1 == 256)) return DGA_E_TILING;
    if ((t.kernelSerial == DGA_KERNEL_STREAMK_TAIL || t.kernelSerial == DGA_KERNEL_STREAMK_ONE_LAUNCH) && !(t.m1 == 256 && t.n1 == 256))
        return DGA_E_TILING;
    return DGA_OK;
}

// the loop-clock build of a variant (dga_gemm_fp8_loop_clock), where one is compiled (DGA_MENU_CLK)
static int (*find_clock_build(const Variant *v, int policy))(const GemmParams &, hipStream_t)
{
    if (v->bm == 256 && v->bn == 256 && policy == DGA_POLICY_CONTINUOUS) return &launch_cfg<GemmCfg<256, 256, 4, 2, 2>, 2, true>;
    if (v->bm == 128 && v->bn == 256 && v->stages == 3 && policy == DGA_POLICY_LOADER_WAVES) return &launch_cfg<GemmCfg<128, 256, 2, 2, 3, 4>, 0, true>;
    return nullptr;
}

// m_indices != nullptr: contiguous-grouped layout -- one A/out matrix of m rows (groups == 1 on that side), b_groups
// B matrices picked per row block by m_indices.  Otherwise b_groups == groups.
// clock_stamps != nullptr (dga_gemm_fp8_loop_clock only): run the loop-clock build of the chosen variant, two words per
// wave go to clock_stamps.
// ix != nullptr: indexed masked-grouped form (a / sfa / out are flat row buffers addressed through ix->row_index).
int run_fp8(const void *a, const float *sfa, const void *b, const float *sfb, void *out,
            const int32_t *masked_m, const int32_t *m_indices, int b_groups, int groups, int m, int n, int k,
            int expected_m, const dga_tiling_t *tiling, void *workspace, size_t workspace_bytes,
            hipStream_t stream, unsigned long long *clock_stamps, const Fp8Indexed *ix, const Fp8Strided *sd)
{
    if (m < 0 || n < 0 || k < 0 || groups < 0 || b_groups < 0) return DGA_E_SHAPE;
    if (groups == 0 || m == 0 || n == 0) return DGA_OK;  // empty problem: nothing to write
    if (!a || !b || !sfa || !sfb || !out) {
        // k == 0 still reads nothing but must write zeros: pointers to out are required
        if (!out || k != 0) return DGA_E_NULL;
    }
    dga_tiling_t local;
    if (!tiling) {
        dga_problem_t pr{};
        pr.m = m; pr.n = n; pr.k = k; pr.groups = groups; pr.expected_m = expected_m;
        pr.layoutTagA = DGA_LAYOUT_ROW_MAJOR; pr.layoutTagB = DGA_LAYOUT_COLUMN_MAJOR;
        pr.layoutTagC = DGA_LAYOUT_ROW_MAJOR; pr.dtype = DGA_DT_FP8_E4M3FN;
        pr.flags = m_indices ? DGA_PROBLEM_CONTIGUOUS_M : 0;
        // A call that names no tiling runs the policy whose outputs stay inside the operator's contract (within 2 bf16 ULP of the
        // fp32-accumulate result: bf16-exact, dispatchPolicyTag 7) unless $DGA_DEFAULT_POLICY says "fast" (the fp8 matrix
        // instruction: twice the rate, product bits ~13 below each octet's largest dropped) or "strict".  $DGA_BF16_EXACT=1
        // (older switch) still forces the bf16-exact pick.
        const int default_policy = dga::default_policy();
        if (default_policy < 0) return DGA_E_RANGE;      // $DGA_DEFAULT_POLICY names no policy: refused, not guessed
        static const int bf16x_auto = [] { const char *e = std::getenv("DGA_BF16_EXACT"); return e ? std::atoi(e) : 0; }();
        int rc = (bf16x_auto || default_policy == 1 || default_policy == 4) ? dga_tiling_bf16_exact(&pr, &local) : dga_tiling(&pr, &local);
        if (rc == DGA_OK && default_policy == 5 && groups == 1 && !m_indices) {   // "auto": bf16-exact where the decode kernel carries it
            dga_tiling_t tb;
            if (dga_tiling_bf16_exact(&pr, &tb) == DGA_OK && tb.kernelSerial == DGA_KERNEL_SPLITK_WORKGROUP) local = tb;
        }
        if (rc == DGA_OK && default_policy == 2) local.dispatchPolicyTag = DGA_POLICY_STRICT;
        if (rc == DGA_OK && (default_policy == 3 || default_policy == 4)) local.dispatchPolicyTag |= DGA_POLICY_UE8M0_SCALES;
        if (rc != DGA_OK) return rc;
        tiling = &local;
    }
    if (int rc = check_tiling(*tiling)) return rc;
    // DGA_POLICY_UE8M0_SCALES: a flag beside the schedule -- the caller promises power-of-two scales; the tile builds that carry
    // the scales in the matrix instruction's E8M0 operands run where they exist (launch_ue8m0), everything else reads the tag
    // without the flag
    dga_tiling_t unflagged;
    bool ue8m0 = false, bx_ue8m0 = false;   // the flag beside a fast-path schedule / beside the bf16-exact policy
    if (tiling->dispatchPolicyTag & DGA_POLICY_UE8M0_SCALES) {
        unflagged = *tiling;
        unflagged.dispatchPolicyTag &= static_cast<uint8_t>(~DGA_POLICY_UE8M0_SCALES);
        ue8m0 = unflagged.dispatchPolicyTag != DGA_POLICY_STRICT && unflagged.dispatchPolicyTag != DGA_POLICY_BF16_EXACT &&
                unflagged.dispatchPolicyTag != DGA_POLICY_PINGPONG;
        bx_ue8m0 = unflagged.dispatchPolicyTag == DGA_POLICY_BF16_EXACT;
        tiling = &unflagged;
    }
    // workspace == NULL is allowed (split-K and the odd-K padding pass are then skipped: single-pass / element-wise
    // kernels, same results); a workspace that is passed must be as large as dga_workspace_bytes() says
    if (workspace && dga_workspace_bytes(tiling) > workspace_bytes) return DGA_E_WORKSPACE;
    // contiguous layout: group segments are aligned to DGA_CONTIGUOUS_M_ALIGNMENT rows, so a tile may not be taller
    // (256-row tiles are legal too: the kernel then runs a second pass on the tiles that straddle two groups)
    if (m_indices && tiling->m1 != 2 * DGA_CONTIGUOUS_M_ALIGNMENT &&
        (tiling->m1 == 0 || tiling->m1 > DGA_CONTIGUOUS_M_ALIGNMENT || DGA_CONTIGUOUS_M_ALIGNMENT % tiling->m1))
        return DGA_E_TILING;
    GemmParams p{};
    p.a = static_cast<const uint8_t *>(a);
    p.sfa = sfa;
    p.b = static_cast<const uint8_t *>(b);
    p.sfb = sfb;
    p.out = static_cast<uint16_t *>(out);
    p.masked_m = masked_m;
    p.m_indices = m_indices;
    p.b_groups = b_groups;
    p.m = m; p.n = n; p.k = k;
    p.kb_n = (k + 127) / 128;
    p.nb_n = (n + 127) / 128;
    p.lda = k; p.ldb = k; p.ldc = n;
    p.a_gs = static_cast<int64_t>(m) * k;
    p.b_gs = static_cast<int64_t>(n) * k;
    p.c_gs = static_cast<int64_t>(m) * n;
    p.sfa_gs = static_cast<int64_t>(m) * p.kb_n;
    p.sfb_gs = static_cast<int64_t>(p.nb_n) * p.kb_n;
    p.sfa_ld = p.kb_n;
    if (sd) {  // dense operands with their own row strides: 16-byte aligned rows are read where they lie (below)
        if (groups != 1 || b_groups != 1 || masked_m || m_indices || ix || clock_stamps) return DGA_E_SHAPE;
        if (sd->lda < k || sd->ldb < k) return DGA_E_SHAPE;
        if ((sd->lda != k && (sd->lda & 15)) || (sd->ldb != k && (sd->ldb & 15))) return DGA_E_ALIGN;
        p.lda = sd->lda; p.ldb = sd->ldb;
    }
    if (ix) {  // one flat source / destination for every group, rows named by the index
        if (!ix->row_index || ix->lda < k || ix->ldc < n || ix->sfa_ld < p.kb_n || ix->rows < 0) return DGA_E_SHAPE;
        p.row_index = ix->row_index;
        p.lda = ix->lda; p.ldc = ix->ldc; p.sfa_ld = ix->sfa_ld;
        p.a_gs = p.c_gs = p.sfa_gs = 0;
        p.a_bytes = ix->rows * ix->lda;
        // the buffer descriptor addresses the source with 32-bit offsets (and marks "beyond K" with bit 31)
        if (p.a_bytes >= 0x7FFFFFFFll) return DGA_E_RANGE;
    }
    p.groups = groups;
    // The weight stream (masked grouped layout; contiguous layout with at most one 128-row block per group on average): every
    // weight byte is read once by one CU and every output row is written once -- the persistent kernel moves both with the
    // non-temporal policy, so that what the L2 retains is the A rows an expert's tiles re-read.  Measured together on
    // 256 x (128, 7168, 2048): full mask 748 -> 698 us, 64 rows 636 -> 590, 16 rows 602 -> 563 (scripts/nt_ab.py; stores
    // alone -4.9 %, loads alone +0.5 % at a full mask).  Dense rasters share their B panels between CUs: never there.
    // $DGA_B_NT (0 / 1 / 2 = by the tile's row count) and $DGA_OUT_NT (0 / 1) override.
    static const int b_nt_env = [] { const char *e = std::getenv("DGA_B_NT"); return e ? std::atoi(e) : -1; }();
    static const int out_nt_env = [] { const char *e = std::getenv("DGA_OUT_NT"); return e ? std::atoi(e) : -1; }();
    const bool weight_stream = groups > 1 ? (masked_m && !m_indices)
                                          : (m_indices && b_groups > 1 && static_cast<int64_t>(m) <= static_cast<int64_t>(b_groups) * DGA_CONTIGUOUS_M_ALIGNMENT);
    p.b_nt = b_nt_env >= 0 ? b_nt_env : (weight_stream ? 1 : 0);
    p.out_nt = out_nt_env >= 0 ? out_nt_env : (weight_stream ? 1 : 0);
    p.splitk = 1;
    p.stamps = clock_stamps;

    // ---- strict policy: the exact-arithmetic kernel takes every shape as it is (no workspace, no padding pass)
    static const int strict_env = [] { const char *e = std::getenv("DGA_STRICT"); return e ? std::atoi(e) : 0; }();
    if (clock_stamps && (tiling->dispatchPolicyTag == DGA_POLICY_STRICT || strict_env || (k % 16) != 0)) return DGA_E_TILING;
    if (tiling->dispatchPolicyTag == DGA_POLICY_STRICT || strict_env) {
        // tile height: 64 rows (two 16-row chains per wave: 105-109 TFLOP/s at 4096^3 where 128 rows -- one wave per SIMD,
        // its two barriers per k block exposed -- reached 91), 32 rows where 64-row tiles would leave CUs idle
        // (128x4096x7168: 647 -> 224 us)
        const int64_t tiles64 = static_cast<int64_t>(groups) * ((m + 63) / 64) * ((n + 127) / 128);
        const int bm = (m > 32 && tiles64 >= static_cast<int64_t>(device_cus())) ? 64 : 32;
        p.tiles_m = (m + bm - 1) / bm;
        p.tiles_n = (n + 127) / 128;
        const int64_t blocks = static_cast<int64_t>(groups) * p.tiles_m * p.tiles_n;
        if (blocks > 0x7FFFFFFFll) return DGA_E_SHAPE;
        const dim3 grid(static_cast<unsigned>(blocks)), block(256);
        // dense problems with at least two 128-row tiles per CU: the 128-row build (fewer conversions and staging permutes per MFMA;
        // 242 VGPRs, two waves per SIMD): 4096^3 1045 -> 1027 us, configs[2] 931 -> 897
        const int64_t tiles128 = static_cast<int64_t>((m + 127) / 128) * ((n + 127) / 128);
        if (groups == 1 && !masked_m && !m_indices && !ix && tiles128 >= 2 * static_cast<int64_t>(device_cus())) {
            p.tiles_m = (m + 127) / 128;
            hipLaunchKernelGGL(gemm_fp8_strict_nt_kernel<4>, dim3(static_cast<unsigned>(tiles128)), block, 0, stream, p);
        } else
        if (bm == 64) hipLaunchKernelGGL(gemm_fp8_strict_nt_kernel<2>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(gemm_fp8_strict_nt_kernel<1>, grid, block, 0, stream, p);
        DGA_HIP_TRY(hipGetLastError());
        return DGA_OK;
    }

    // ---- workspace carve: [padded A | padded B] (K % 16 != 0), then [split-K slabs]
    uint8_t *ws = static_cast<uint8_t *>(workspace);
    size_t ws_used = 0;
    auto carve = [&](size_t bytes) -> uint8_t * {
        const size_t at = (ws_used + 255) & ~size_t(255);
        if (!ws || at + bytes > workspace_bytes) return nullptr;
        ws_used = at + bytes;
        return ws + at;
    };
    // ---- PaddingCommon (kernelSerial 2): K % 16 != 0 WITHOUT the padded copies -- the loader waves of the 128 x 256 tile fetch
    //      the rows where they lie (any byte alignment), realign them in registers and write the LDS image themselves (the
    //      reference's kernel of that name fuses its re-layout with the matmul the same way:
    //      op_kernel/kernel/padding_common_matmul_kernel.h:33-107).  Dense, fp8 matrix instruction; anything it does not take
    //      (DGA_E_TILING) goes on to the padding pass below.  It measured 15-40 % slower than padding pass + aligned tile
    //      (profiles/r04_odd_k_fused.txt: a misaligned 128-byte row piece costs two line requests on every re-read), so the
    //      selector never asks for it; it runs when the tiling names it, when $DGA_UNALIGNED = 1, or when the caller gave no
    //      workspace for the padded copies (the alternative there is the element-wise kernel, orders of magnitude slower).
    static const int unal_env = [] { const char *e = std::getenv("DGA_UNALIGNED"); return e ? std::atoi(e) : -1; }();
    static const int bf16x_env0 = [] { const char *e = std::getenv("DGA_BF16_EXACT"); return e ? std::atoi(e) : 0; }();
    if (k > 0 && (k % 16) != 0 && !ix && !sd && groups == 1 && !masked_m && !m_indices && !clock_stamps && !bf16x_env0 &&
        tiling->dispatchPolicyTag != DGA_POLICY_BF16_EXACT &&
        (unal_env >= 0 ? unal_env != 0
                       : (tiling->kernelSerial == DGA_KERNEL_PADDING_COMMON ||
                          // no room for the padded copies: in place through the loader waves instead of the element-wise kernel
                          !workspace || workspace_bytes < (((static_cast<size_t>(m) * p.kb_n * 128 + 255) & ~size_t(255)) +
                                                           static_cast<size_t>(n) * p.kb_n * 128 + 256)))) {
        GemmParams q = p;
        q.tiles_m = (m + 127) / 128;
        q.tiles_n = (n + 255) / 256;
        q.raster_group = tiling->swizzleOffset ? tiling->swizzleOffset : 1;
        static const int xcd_remap_u = [] { const char *e = std::getenv("DGA_XCD_REMAP"); return e ? std::atoi(e) : 1; }();
        q.xcd_remap = xcd_remap_u;
        const int rc = launch_unaligned(q, stream);
        if (rc != DGA_E_TILING) return rc;
    }
    if (sd && k > 0) {
        // Row-strided operands: an operand whose rows start on 16-byte boundaries, are at least round_up(K, 16) bytes apart and --
        // when K % 16 != 0 -- carry zeros from byte K to that boundary (DGA_ROWS_*_ZERO_PADDED: the caller's promise; the
        // quantisers' _ld forms write them) is read in place; the other one, if any, goes through the padding pass alone.
        const int k16 = (k + 15) & ~15;
        auto in_place = [&](const void *base, int64_t ld, int flag) {
            return (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (ld & 15) == 0 && ld >= k16 && ((k % 16) == 0 || (sd->flags & flag));
        };
        const bool a_ok = in_place(p.a, p.lda, DGA_ROWS_A_ZERO_PADDED), b_ok = in_place(p.b, p.ldb, DGA_ROWS_B_ZERO_PADDED);
        bool ready = a_ok && b_ok;
        if (!ready) {
            const int kp = p.kb_n * 128;
            uint8_t *pa = a_ok ? nullptr : carve(static_cast<size_t>(m) * kp);
            uint8_t *pb = b_ok ? nullptr : carve(static_cast<size_t>(n) * kp);
            if ((a_ok || pa) && (b_ok || pb)) {
                const int st = pad_rows_strided(a_ok ? nullptr : p.a, p.lda, pa, a_ok ? 0 : m, b_ok ? nullptr : p.b, p.ldb, pb,
                                                b_ok ? 0 : n, k, kp, stream);
                if (st != DGA_OK) return st;
                if (!a_ok) { p.a = pa; p.lda = kp; }
                if (!b_ok) { p.b = pb; p.ldb = kp; }
                ready = true;
            }
            // no (or too small a) workspace: the element-wise kernel below still computes the right answer
        }
        if (ready) { p.k = k16; k = k16; }   // (the tile kernels zero-fill from there to the end of the last k block)
    } else
    if (k > 0 && (k % 16) != 0 && !ix) {   // (indexed rows are read where they lie: odd K takes the element-wise kernel)
        const int kp = p.kb_n * 128;
        const int64_t rows_a = static_cast<int64_t>(groups) * m, rows_b = static_cast<int64_t>(b_groups) * n;
        uint8_t *pa = carve(static_cast<size_t>(rows_a) * kp);
        uint8_t *pb = pa ? carve(static_cast<size_t>(rows_b) * kp) : nullptr;
        if (pa && pb && ((reinterpret_cast<uintptr_t>(pa) | reinterpret_cast<uintptr_t>(pb)) & 15) == 0) {
            const int st = pad_rows(p.a, pa, rows_a, p.b, pb, rows_b, k, kp, stream);   // both operands, one launch
            if (st != DGA_OK) return st;
            p.a = pa; p.b = pb;
            p.k = kp; p.lda = kp; p.ldb = kp;
            p.a_gs = static_cast<int64_t>(m) * kp;
            p.b_gs = static_cast<int64_t>(n) * kp;
            k = kp;  // the padded operands are what the tile kernel sees (scales and k-block count are unchanged)
        }
        // no (or too small a) workspace: the element-wise kernel below still computes the right answer
    }

    // LDS-DMA kernel: 16-byte chunks (K % 16 == 0, 16-byte aligned bases) and 32-bit in-tile byte offsets
    const bool fast_ok = (k % 16 == 0) && k > 0 && ((reinterpret_cast<uintptr_t>(p.a) & 15) == 0) &&
                         ((reinterpret_cast<uintptr_t>(p.b) & 15) == 0) && (p.lda % 16 == 0) && (p.ldb % 16 == 0) &&
                         (static_cast<int64_t>(p.lda) * 257 < 0x7FFFFFFFll) && (static_cast<int64_t>(p.ldb) * 257 < 0x7FFFFFFFll);
    if (!fast_ok) {
        // K not a multiple of the 16-byte DMA chunk and no workspace to pad into (or k == 0): element-wise kernel
        dim3 grid((n + 15) / 16, (m + 15) / 16, groups);
        hipLaunchKernelGGL(gemm_fp8_blockscaled_nt_generic_kernel, grid, dim3(256), 0, stream, p);
        DGA_HIP_TRY(hipGetLastError());
        return DGA_OK;
    }
    static const int bf16x_env = [] { const char *e = std::getenv("DGA_BF16_EXACT"); return e ? std::atoi(e) : 0; }();
    const bool bf16x = tiling->dispatchPolicyTag == DGA_POLICY_BF16_EXACT || bf16x_env;
    const Bf16xVariant *vx = bf16x ? find_bf16x_variant(tiling->m1, tiling->n1) : nullptr;
    if (bf16x && (!vx || clock_stamps)) return DGA_E_TILING;
    // the 128 x 256 tile's image builds (bf16 LDS image converted once per workgroup; same bits as the in-register build; dense
    // and masked-grouped layouts).  Both-operand images measured 10 % SLOWER than the in-register build, the A-only image ties it
    // (profiles/r04_bximg_stamps.txt, r04_aimage.txt), so they run only when NAMED: tiling.build = DGA_BUILD_BX_AIMAGE (A image), _IMAGE8 (both operands, 8
    // waves), 6 (both operands, 4 waves), or $DGA_BX_IMAGE = 1 / 8 / 4 (0: never).  (Stage counts no tile build has, so that a fast-path
    // tiling that is handed this policy's tag -- 2 x 2 waves, two stages -- cannot name one by accident: it did, for a while, on every
    // grouped call of the policy.)
    static const int bx_image_env = [] { const char *e = std::getenv("DGA_BX_IMAGE"); return e ? std::atoi(e) : -1; }();
    int bx_image = 0;
    if (vx && vx->bm == 128 && vx->bn == 256 && !m_indices && !ix) {
        if (bx_image_env >= 0) bx_image = bx_image_env == 4 ? 4 : (bx_image_env == 1 ? 1 : (bx_image_env ? 8 : 0));
        else if (tiling->build == DGA_BUILD_BX_AIMAGE) bx_image = 1;
        else if (tiling->build == DGA_BUILD_BX_IMAGE8) bx_image = 8;
        else if (tiling->build == DGA_BUILD_BX_IMAGE4) bx_image = 4;
    }
    const Variant *v = find_variant(tiling->m1, tiling->n1, tiling->wavesM, tiling->wavesN, tiling->stages);
    if (!v && !vx) return DGA_E_TILING;
    const int tile_m = vx ? vx->bm : v->bm, tile_n = vx ? vx->bn : v->bn;
    const bool wants_loaders = tiling->dispatchPolicyTag == DGA_POLICY_LOADER_WAVES || tiling->dispatchPolicyTag == DGA_POLICY_PERSISTENT;
    if (!vx && wants_loaders && !v->launch_lc)   // the tile's build that has loader waves
        for (int i = 0; i < kNumVariants; ++i)
            if (kVariants[i].bm == v->bm && kVariants[i].bn == v->bn && kVariants[i].stages == v->stages && kVariants[i].launch_lc) {
                v = &kVariants[i];
                break;
            }
    p.tiles_m = (m + tile_m - 1) / tile_m;
    p.tiles_n = (n + tile_n - 1) / tile_n;
    // Dense rasters of at most two rounds: every CU stores its tile at the same moment, and rows written "sc0 sc1" (write-through)
    // do not wait in the XCD's L2 for the kernel-end write-back -- 4096^3: fast 60.3 -> 58.9 us, hardware-scale 55.6 -> 54.1,
    // bf16-exact 121.3 -> 119.3; the nt policy is 2 % SLOWER there, longer rasters are level (profiles/r05_out_store_policy.txt)
    if (out_nt_env < 0 && groups == 1 && !masked_m && !m_indices && !ix &&
        static_cast<int64_t>(p.tiles_m) * p.tiles_n <= 2 * static_cast<int64_t>(device_cus()))
        p.out_nt = 2;
    p.raster_group = tiling->swizzleOffset ? tiling->swizzleOffset : 1;
    static const int xcd_remap = [] { const char *e = std::getenv("DGA_XCD_REMAP"); return e ? std::atoi(e) : 1; }();
    p.xcd_remap = xcd_remap;

    // ---- workgroup split-K (kernelSerial 6): one launch, the K slices are the waves of a workgroup; dense, M <= 64.  fp8 matrix
    //      instruction only; a shape it does not take (DGA_E_TILING) falls through to the tiling's tile kernel
    static const int wsk_env = [] { const char *e = std::getenv("DGA_WSK"); return e ? std::atoi(e) : -1; }();
    // ... and, under the bf16-exact policy, its form for up to a few 64-row tiles (tiling.build = DGA_BUILD_BX_DECODE;
    // gemm_fp8_bf16x_dsk_kernel.hpp): two k groups per workgroup, splitkFactor workgroups per tile through the workspace.  What it does
    // not take runs the two-launch split-K of the same tiling below.
    if (bf16x && !bx_ue8m0 && !clock_stamps && groups == 1 && !masked_m && !m_indices && !ix && tiling->kernelSerial == DGA_KERNEL_SPLITK_WORKGROUP &&
        tiling->build == DGA_BUILD_BX_DECODE) {
        const int64_t dsk_tiles = static_cast<int64_t>((m + 63) / 64) * ((n + 127) / 128);
        const int s = bx_dsk_splits(dsk_tiles, p.kb_n, tiling->splitkFactor, static_cast<int>(device_cus()));
        const size_t need = s > 0 ? bx_dsk_workspace_bytes(dsk_tiles, s) : 0;
        uint8_t *dsk_ws = need ? carve(need) : nullptr;
        if (s > 0 && (need == 0 || dsk_ws)) {
            const int rc = launch_bf16x_dsk(p, s, dsk_ws, need, stream);
            if (rc != DGA_E_TILING) return rc;
        }
    }
    if (bf16x && !clock_stamps && groups == 1 && !masked_m && !m_indices && !ix && tiling->build != DGA_BUILD_BX_DECODE &&
        (wsk_env >= 0 ? wsk_env != 0 : tiling->kernelSerial == DGA_KERNEL_SPLITK_WORKGROUP)) {
        const int rc = launch_wsk_dma(p, stream, 1);   // the bf16-exact policy has the LDS-DMA build only (M <= 32)
        if (rc != DGA_E_TILING) return rc;
    }
    if (!bf16x && !clock_stamps && groups == 1 && !masked_m && !m_indices && !ix &&
        (wsk_env >= 0 ? wsk_env != 0 : tiling->kernelSerial == DGA_KERNEL_SPLITK_WORKGROUP)) {
        // M <= 32: the LDS-DMA staged build (whole-line requests, hand-counted vmcnt); a tiling with build = DGA_BUILD_WSK_REGISTER names the register
        // build (fragments global -> registers), which also takes 32 < M <= 64.  $DGA_WSK_DMA = 0 / 1 overrides.
        static const int wsk_dma_env = [] { const char *e = std::getenv("DGA_WSK_DMA"); return e ? std::atoi(e) : -1; }();
        if (wsk_dma_env >= 0 ? wsk_dma_env != 0 : tiling->build != DGA_BUILD_WSK_REGISTER) {
            const int rc = launch_wsk_dma(p, stream);
            if (rc != DGA_E_TILING) return rc;
        }
        const int rc = launch_wsk(p, stream);
        if (rc != DGA_E_TILING) return rc;
    }

    // ---- split-K (kernelSerial 4): partial fp32 slabs + combine; dense only
    if (clock_stamps && (tiling->splitkFactor > 1 || tiling->kernelSerial == DGA_KERNEL_STREAMK_TAIL)) return DGA_E_TILING;
    if (tiling->splitkFactor > 1 && groups == 1 && !masked_m && !m_indices) {
        int s = tiling->splitkFactor;
        const int kbps = (p.kb_n + s - 1) / s;
        s = (p.kb_n + kbps - 1) / kbps;  // no empty split
        float *slabs = s > 1 ? reinterpret_cast<float *>(carve(static_cast<size_t>(s) * m * n * sizeof(float))) : nullptr;
        if (slabs) {
            p.splitk = s;
            p.kb_per_split = kbps;
            p.partial = slabs;
            GemmParams pk = p;
            pk.groups = s;  // grid = splitk x tiles
            int rc = DGA_E_TILING;
            if (vx && bx_ue8m0 && !bx_image) rc = launch_bf16u(vx->bm, vx->bn, pk, stream);
            if (rc == DGA_E_TILING)
                rc = vx ? (bx_image ? launch_bf16x_image(pk, bx_image, stream) : vx->launch(pk, stream)) : v->launch(pk, stream);
            if (rc != DGA_OK) return rc;
            const int64_t mn = static_cast<int64_t>(m) * n;
            hipLaunchKernelGGL(splitk_reduce_bf16_kernel, dim3(static_cast<unsigned>((mn / 8 + 255) / 256 + 1)), dim3(256),
                               0, stream, slabs, p.out, mn, s);
            DGA_HIP_TRY(hipGetLastError());
            return DGA_OK;
        }
        // workspace missing: fall through to the single-pass kernel (same result, fewer CUs busy)
    }
    static const int pp_env = [] { const char *e = std::getenv("DGA_PINGPONG"); return e ? std::atoi(e) : -1; }();
    const int policy = pp_env >= 0 ? pp_env : tiling->dispatchPolicyTag;
    if (vx) {   // bf16-exact: one launch over the whole raster
        if (bx_ue8m0 && !bx_image && !clock_stamps && tiling->kernelSerial != DGA_KERNEL_STREAMK_TAIL) {   // power-of-two scales: folded into the conversions, the MFMA accumulates in place
            const int rc = launch_bf16u(vx->bm, vx->bn, p, stream);
            if (rc != DGA_E_TILING) return rc;
        }
        if (bx_image) return launch_bf16x_image(p, bx_image, stream);
        // tail in quarter tiles (kernelSerial 5), as on the fast path below: the whole rounds of 128 x 256 tiles run as they are, the
        // last partial round -- at most half of the CUs' worth of tiles -- is covered by 64 x 128 tiles (four per parent tile, two
        // workgroups to a CU) in a second launch.  Same arithmetic in the same k order: the bytes are those of one launch.
        if (tiling->kernelSerial == DGA_KERNEL_STREAMK_TAIL && groups == 1 && !masked_m && !m_indices && !ix && vx->bm == 128 && vx->bn == 256) {
            const int tiles = p.tiles_m * p.tiles_n, cus = static_cast<int>(device_cus());
            const int tail = tiles % cus, main_tiles = tiles - tail;
            const Bf16xVariant *vq = find_bf16x_variant(64, 128);
            if (tail > 0 && tail * 2 <= cus && main_tiles > 0 && vq && vq->bm == 64 && vq->bn == 128) {
                GemmParams pm = p;
                pm.launch_tiles = main_tiles;
                int rc = launch_bf16x_persistent(pm, stream);    // (DGA_E_TILING: a launch it does not take -- one k block)
                if (rc == DGA_E_TILING) rc = vx->launch(pm, stream);
                if (rc != DGA_OK) return rc;
                GemmParams pt = p;  // tiles_m / tiles_n / raster_group stay those of the parent raster
                pt.tail_begin = main_tiles;
                pt.tail_sub = 2;
                pt.launch_tiles = tail * 4;
                return vq->launch(pt, stream);
            }
        }
        // the 128 x 256 tile's persistent form (gemm_fp8_bf16x_persistent_kernel.hpp; same bits).  The dispatcher hides most of a tile
        // boundary already, so it pays little -- masked grouped 256 x (128, 7168, 2048): full mask 1010 -> 998 us, random masks
        // 856 -> 835; 4096^3 114.1 -> 113.4; configs[2], 1.75 tiles per CU, 97.3 -> 98.5 (profiles/r04_bf16x_persistent_ab.txt) --
        // and runs on every masked grouped or dense raster of more than one round: uneven rasters gain most where K is short -- 4096 x 7168 x
        // 2048 (3.5 rounds) 121.8 -> 111.5 us, 8064 x 4096 x 512 48.3 -> 40.0, 6016 x 4096 x 4096 173.8 -> 167.7; at exactly one round it
        // loses 1.6 % (profiles/r05_bx_persist_ab.txt).  tiling.build = DGA_BUILD_BX_PERSISTENT names it, _ONE_TILE the one-tile build, $DGA_BF16X_PERSIST = 0 / 1
        // overrides.
        static const int bxp_env = [] { const char *e = std::getenv("DGA_BF16X_PERSIST"); return e ? std::atoi(e) : -1; }();
        const int64_t tiles = static_cast<int64_t>(p.groups) * p.tiles_m * p.tiles_n, cus = static_cast<int64_t>(device_cus());
        // Stream-K in one launch (kernelSerial 7 under this policy; gemm_fp8_bf16x_streamk_kernel.hpp): whole rounds as the persistent kernel
        // runs them, the last partial round cut along K with fp32 partial tiles through the workspace.  What it does not take (no partial
        // round, no workspace, a CU mask) runs the builds below.
        if (tiling->kernelSerial == DGA_KERNEL_STREAMK_ONE_LAUNCH && vx->bm == 128 && vx->bn == 256 && !clock_stamps && groups == 1 &&
            !masked_m && !m_indices && !ix) {
            const size_t need = bx_streamk_workspace_bytes();
            if (uint8_t *sk_ws = carve(need)) {
                const int rc = launch_bf16x_streamk(p, sk_ws, need, stream);
                if (rc != DGA_E_TILING) return rc;
            }
        }
        // the masked grouped layout's own kernel (gemm_fp8_bf16x_grouped_kernel.hpp; same bits): two k blocks of the ring in flight and
        // the loop unrolled for the m-tiles that hold rows.  tiling.build = DGA_BUILD_BX_GROUPED names it (dga_tiling_bf16_exact does), $DGA_BX_GROUPED = 0 / 1 overrides.
        static const int bxg_env = [] { const char *e = std::getenv("DGA_BX_GROUPED"); return e ? std::atoi(e) : -1; }();
        // (a dense raster runs it too when the tiling names it: the loop is the same, every tile has all its rows)
        if (vx->bm == 128 && vx->bn == 256 && !clock_stamps && !m_indices &&
            (bxg_env >= 0 ? (bxg_env != 0 && masked_m) : tiling->build == DGA_BUILD_BX_GROUPED)) {
            const int rc = launch_bf16x_grouped(p, stream);
            if (rc != DGA_E_TILING) return rc;
        }
        const bool pays = tiles > cus;
        if (vx->bm == 128 && vx->bn == 256 && !clock_stamps &&
            (bxp_env >= 0 ? bxp_env != 0 : (tiling->build == DGA_BUILD_BX_PERSISTENT || (tiling->build != DGA_BUILD_BX_ONE_TILE && pays)))) {
            const int rc = launch_bf16x_persistent(p, stream);
            if (rc != DGA_E_TILING) return rc;
        }
        return vx->launch(p, stream);
    }
    auto launch_main = [&](const GemmParams &q) -> int {
        // (the persistent loader-wave form has no hardware-scale build: on the grouped weight stream -- bound by HBM, not by the
        //  vector pipe -- the non-persistent one measured 16 % slower than the persistent promotion form, 807 against 697 us)
        const bool runs_persistent = policy == DGA_POLICY_PERSISTENT && v->launch_ps && q.splitk <= 1 && !q.tail_sub && q.launch_tiles == 0;
        if (ue8m0 && !q.stamps && !runs_persistent) {   // power-of-two scales: the build that accumulates in the MFMA, where the tile has one
            const bool cont = (policy == DGA_POLICY_CONTINUOUS || policy == DGA_POLICY_CONTINUOUS_PERSISTENT) && v->launch_cont;
            const bool loaders = (policy == DGA_POLICY_LOADER_WAVES || policy == DGA_POLICY_PERSISTENT) && v->launch_lc;
            if (cont && v->bm == 256 && v->bn == 256 && tiling->wavesM == 2 && tiling->wavesN == 2 && !q.tail_sub) {
                const int r4 = launch_ue8m0_w4(q, stream);
                if (r4 != DGA_E_TILING) return r4;
            }
            const int rc = launch_ue8m0(v->bm, v->bn, loaders, cont, q, stream);
            if (rc != DGA_E_TILING) return rc;
        }
        if (q.stamps) {
            auto clk = find_clock_build(v, (policy == 2 || policy == DGA_POLICY_CONTINUOUS_PERSISTENT) && v->launch_cont ? 2 : (policy == DGA_POLICY_LOADER_WAVES && v->launch_lc ? policy : 0));
            return clk ? clk(q, stream) : DGA_E_TILING;
        }
        if (policy == DGA_POLICY_PERSISTENT && v->launch_ps && q.splitk <= 1 && !q.tail_sub && q.launch_tiles == 0)
            return v->launch_ps(q, stream);
        if ((policy == DGA_POLICY_LOADER_WAVES || policy == DGA_POLICY_PERSISTENT) && v->launch_lc) return v->launch_lc(q, stream);
        if (policy == 1 && v->launch_pp) return v->launch_pp(q, stream);
        if (policy == DGA_POLICY_CONTINUOUS_PERSISTENT && v->bm == 256 && v->bn == 256 && v->launch_cont) {
            const int rc = launch_cont_persistent(q, stream);    // DGA_E_TILING = not a raster it takes: the one-tile build
            return rc == DGA_E_TILING ? v->launch_cont(q, stream) : rc;
        }
        if ((policy == 2 || policy == DGA_POLICY_CONTINUOUS_PERSISTENT) && v->launch_cont) return v->launch_cont(q, stream);
        return v->launch(q, stream);
    };

    // ---- Stream-K proper (kernelSerial 7): one launch, the raster's k blocks cut evenly over the CUs, fp32 partial tiles through the
    //      workspace (gemm_fp8_streamk_kernel.hpp).  What it does not take (ragged shapes, no workspace) runs the tile kernel below.
    if (tiling->kernelSerial == DGA_KERNEL_STREAMK_ONE_LAUNCH && !vx && !clock_stamps && groups == 1 && !masked_m && !m_indices && !ix &&
        v->bm == 256 && v->bn == 256) {
        const size_t need = streamk_workspace_bytes();
        if (uint8_t *sk_ws = carve(need)) {
            const int rc = launch_streamk(p, sk_ws, need, ue8m0, stream);
            if (rc != DGA_E_TILING) return rc;
        }
    }

    // ---- tail in quarter tiles (kernelSerial 5): the whole waves of 256x256 tiles run as they are; the last partial wave
    //      is covered by 128x128 tiles (four per parent tile) in a second launch, so that it occupies four times as many
    //      CUs for a fraction of a round -- the purpose of the reference's Stream-K handler (select_kernel.cpp:303-331,
    //      wave quantisation) without partial sums: every output still comes from one accumulation in k order, the
    //      bytes are those of a single launch.
    if (tiling->kernelSerial == DGA_KERNEL_STREAMK_TAIL && groups == 1 && !masked_m && !m_indices && v->bm == 256 && v->bn == 256) {
        const int tiles = p.tiles_m * p.tiles_n, cus = static_cast<int>(device_cus());
        const int tail = tiles % cus, main_tiles = tiles - tail;
        const Variant *vq = find_variant(128, 128, 0, 0, 3);
        // (up to half of the CUs' worth of parent tiles: two quarter tiles per CU.  A tail longer than a quarter of the RASTER used to
        //  fault -- the kernel read a quarter-tile index beyond the parent tile count as a group index; fixed there, tested in
        //  tests/test_gemm_gpu.py)
        if (tail > 0 && tail * 2 <= cus && main_tiles > 0 && vq) {
            GemmParams pm = p;
            pm.launch_tiles = main_tiles;
            int rc = launch_main(pm);
            if (rc != DGA_OK) return rc;
            GemmParams pt = p;  // tiles_m / tiles_n / raster_group stay those of the parent raster
            pt.tail_begin = main_tiles;
            pt.tail_sub = 2;
            pt.launch_tiles = tail * 4;
            // (the quarter tiles give a CU at most two workgroups: the 4 + 4 loader-wave build of the tile, -1..-2 % on the whole call:
            //  1024 x 18432 x 7168 139.8 -> 136.9 us, 4608 x 4096 x 7168 139.2 -> 136.2; same arithmetic, same bytes)
            if (ue8m0) {
                const int rq = launch_ue8m0(128, 128, vq->launch_lc != nullptr, false, pt, stream);
                if (rq != DGA_E_TILING) return rq;
            }
            if (vq->launch_lc) return vq->launch_lc(pt, stream);
            return vq->launch(pt, stream);
        }
    }
    return launch_main(p);
}

}  // namespace dga

extern "C" {

int dga_default_policy(char *name_out, int cap)
{
    const int v = dga::default_policy();
    if (v < 0) return DGA_E_RANGE;
    if (name_out && cap > 0) {
        std::strncpy(name_out, dga::kPolicyNames[v], static_cast<size_t>(cap) - 1);
        name_out[cap - 1] = 0;
    }
    return DGA_OK;
}

int dga_tiling_check(const dga_tiling_t *tiling)
{
    if (!tiling) return DGA_E_NULL;
    return dga::check_tiling(*tiling);
}

int dga_gemm_fp8_fp8_bf16_nt(const void *a, const float *sfa, const void *b, const float *sfb, void *out, int m,
                             int n, int k, const dga_tiling_t *tiling, void *workspace, size_t workspace_bytes,
                             void *stream)
{
    return dga::run_fp8(a, sfa, b, sfb, out, nullptr, nullptr, 1, 1, m, n, k, 0, tiling, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream), nullptr, nullptr);
}

int dga_gemm_fp8_fp8_bf16_nt_strided(const void *a, int64_t lda, const float *sfa, const void *b, int64_t ldb, const float *sfb,
                                     void *out, int m, int n, int k, int flags, const dga_tiling_t *tiling, void *workspace,
                                     size_t workspace_bytes, void *stream)
{
    const dga::Fp8Strided sd{lda, ldb, flags};
    return dga::run_fp8(a, sfa, b, sfb, out, nullptr, nullptr, 1, 1, m, n, k, 0, tiling, workspace, workspace_bytes,
                        static_cast<hipStream_t>(stream), nullptr, nullptr, &sd);
}

int dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked(const void *a, const float *sfa, const void *b, const float *sfb,
                                              void *out, const int32_t *masked_m, int groups, int m_max, int n,
                                              int k, int expected_m, const dga_tiling_t *tiling, void *workspace,
                                              size_t workspace_bytes, void *stream)
{
    if (groups > 0 && m_max > 0 && !masked_m) return DGA_E_NULL;
    return dga::run_fp8(a, sfa, b, sfb, out, masked_m, nullptr, groups, groups, m_max, n, k, expected_m, tiling,
                        workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr);
}

int dga_m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(const void *a, const float *sfa, const void *b, const float *sfb,
                                                  void *out, const int32_t *m_indices, int m_sum, int groups, int n,
                                                  int k, const dga_tiling_t *tiling, void *workspace,
                                                  size_t workspace_bytes, void *stream)
{
    if (groups < 0) return DGA_E_SHAPE;
    if (m_sum > 0 && n > 0 && groups > 0 && !m_indices) return DGA_E_NULL;
    if (groups == 0) return DGA_OK;  // no B matrices: every row is a padding row
    return dga::run_fp8(a, sfa, b, sfb, out, nullptr, m_indices, groups, 1, m_sum, n, k, 0, tiling, workspace,
                        workspace_bytes, static_cast<hipStream_t>(stream), nullptr, nullptr);
}

int dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(const void *a, int64_t lda, const float *sfa, int64_t sfa_ld,
                                                      const void *b, const float *sfb, void *out, int64_t ldc,
                                                      const int64_t *row_index, int64_t rows, const int32_t *masked_m,
                                                      int groups, int m_max, int n, int k, int expected_m,
                                                      const dga_tiling_t *tiling, void *workspace, size_t workspace_bytes,
                                                      void *stream)
{
    if (groups > 0 && m_max > 0 && (!masked_m || !row_index)) return DGA_E_NULL;
    const dga::Fp8Indexed ix{row_index, lda, sfa_ld, ldc, rows};
    return dga::run_fp8(a, sfa, b, sfb, out, masked_m, nullptr, groups, groups, m_max, n, k, expected_m, tiling,
                        workspace, workspace_bytes, static_cast<hipStream_t>(stream), nullptr, &ix);
}

int dga_last_hip_error(void) { return dga::g_last_hip_error.load(); }

int dga_device_platform(dga_platform_t *out)
{
    if (!out) return DGA_E_NULL;
    dga_platform_mi355x(out);
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return DGA_E_HIP;
    out->coreNum = static_cast<uint32_t>(prop.multiProcessorCount);
    out->l1Size = prop.maxSharedMemoryPerMultiProcessor ? prop.maxSharedMemoryPerMultiProcessor : out->l1Size;
    out->waveSize = static_cast<uint32_t>(prop.warpSize);
    return DGA_OK;
}

}  // extern "C"
