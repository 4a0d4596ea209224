// fp8 tile-kernel menu, part M: the one-launch Stream-K build of the bf16-exact policy's persistent 128 x 256 kernel
// (gemm_fp8_bf16x_streamk_kernel.hpp; dispatchPolicyTag 7 with kernelSerial DGA_KERNEL_STREAMK_ONE_LAUNCH): whole rounds as they are,
// the last partial round cut along K, fp32 partial tiles through the caller's workspace.
#include <atomic>
#include <cstdlib>
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_bf16x_streamk_kernel.hpp"
namespace dga {

// Workgroups a launch on `stream` can count on being resident TOGETHER, one per CU: the device's CUs unless something narrows the
// queue's CU set -- $HSA_CU_MASK / $ROC_GLOBAL_CU_MASK (process-wide), a stream created with hipExtStreamCreateWithCUMask.  Then 0:
// a kernel whose workgroups wait for one another must not be launched (the waiting ones would hold every usable CU while the rest
// never start); the callers run the tile kernels instead.
int coresident_workgroups(hipStream_t stream)
{
    static const bool env_mask = [] {
        for (const char *name : {"HSA_CU_MASK", "ROC_GLOBAL_CU_MASK"}) {
            const char *e = std::getenv(name);
            if (e && *e) return true;
        }
        return false;
    }();
    if (env_mask) return 0;
    const int cus = static_cast<int>(device_cus());
    uint32_t mask[16] = {0};
    if (stream && hipExtStreamGetCUMask(stream, 16, mask) == hipSuccess) {
        int bits = 0;
        for (uint32_t w : mask) bits += __builtin_popcount(w);
        if (bits > 0 && bits < cus) return 0;
    } else {
        (void)hipGetLastError();   // (the null stream / a runtime without the query: the device's CUs)
    }
    return cus;
}

size_t bx_streamk_workspace_bytes() { return static_cast<size_t>(device_cus()) * (128 * 256 * 4 + 8) + 256; }

template <bool KTAIL>
static int launch_bx_streamk_one(const GemmParams &p, const StreamKArgs &sk, unsigned grid, hipStream_t stream)
{
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    auto kfn = gemm_fp8_bf16x_streamk_kernel<KTAIL>;
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, sk);
    return record_hip(hipGetLastError());
}

// ws: the caller's workspace, at least bx_streamk_workspace_bytes().  DGA_E_TILING: not a launch this kernel takes (no partial round to
// cut, a layout it does not have, co-residency not guaranteed, no or too small a workspace): the caller runs the tiling's tile kernel.
int launch_bf16x_streamk(const GemmParams &p, void *ws, size_t ws_bytes, hipStream_t stream)
{
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.splitk > 1 || p.tail_sub || p.stamps || p.launch_tiles) return DGA_E_TILING;
    if (p.kb_n < 2) return DGA_E_TILING;
    const int grid = coresident_workgroups(stream);
    if (grid <= 0) return DGA_E_TILING;
    const int64_t tiles = static_cast<int64_t>(p.tiles_m) * p.tiles_n;
    if (tiles <= 0 || tiles > 0x3FFFFFFF) return DGA_E_TILING;
    const BxStreamKPlan pl = bx_streamk_plan(static_cast<int>(tiles), grid, p.kb_n);
    if (pl.form == 0) return DGA_E_TILING;     // nothing to cut: the persistent kernel
    if (!ws || ws_bytes < static_cast<size_t>(grid) * (128 * 256 * 4 + 8) || (reinterpret_cast<uintptr_t>(ws) & 15)) return DGA_E_TILING;
    StreamKArgs sk;
    sk.partials = static_cast<float *>(ws);
    sk.flags = reinterpret_cast<unsigned long long *>(sk.partials + static_cast<size_t>(grid) * (128 * 256));
    // A flag is raised when it holds this launch's epoch: 64 mixed bits no earlier launch used and stale workspace bytes will not
    // hold -- nothing to zero.  Its reader puts it back to 0, so a launch that is being CAPTURED into a graph (replayed with the same
    // arguments) needs no memset node either: every replay finds zeros.
    static std::atomic<unsigned long long> launches{0};
    const unsigned long long e = launches.fetch_add(1) + 1;
    sk.epoch = (e * 0x9E3779B97F4A7C15ull) | 1ull;
#ifdef DGA_BXSK_KNOBS
    GemmParams q = p;
    if (const char *e = std::getenv("DGA_BXSK_KNOB")) q.tail_begin = std::atoi(e);
    return (p.k % 128) ? launch_bx_streamk_one<true>(q, sk, static_cast<unsigned>(grid), stream)
                       : launch_bx_streamk_one<false>(q, sk, static_cast<unsigned>(grid), stream);
#endif
    return (p.k % 128) ? launch_bx_streamk_one<true>(p, sk, static_cast<unsigned>(grid), stream)
                       : launch_bx_streamk_one<false>(p, sk, static_cast<unsigned>(grid), stream);
}

}  // namespace dga
