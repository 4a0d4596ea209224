// fp8 tile-kernel menu, part L: the bf16-exact policy's kernel for the masked grouped layout (gemm_fp8_bf16x_grouped_kernel.hpp,
// dispatchPolicyTag 7): persistent, two k blocks of the ring in flight, the loop unrolled for the m-tiles of a wave that hold rows.
#include <cstdlib>
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_bf16x_grouped_kernel.hpp"
namespace dga {

template <bool KTAIL, bool BNT, bool STAG, bool IDX = false>
static int launch_bf16x_grouped_one(const GemmParams &p, hipStream_t stream)
{
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    auto kfn = gemm_fp8_bf16x_grouped_kernel<KTAIL, BNT, STAG, IDX>;
    constexpr int kLds = Cfg::LDS_BYTES + (IDX ? 256 * 16 : 0);      // the indexed build keeps 16 bytes per thread of waves 4..7 behind the ring
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    const int64_t tiles = static_cast<int64_t>(p.groups) * p.tiles_m * p.tiles_n;
    if (tiles == 0) return DGA_OK;
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(tiles, device_cus()));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), kLds, stream, p);
    return record_hip(hipGetLastError());
}

int launch_bf16x_grouped(const GemmParams &p, hipStream_t stream)
{
    // whole masked-grouped (or dense) rasters of at least two k blocks, packed or indexed rows; split-K, quarter tiles, part launches and
    // the contiguous layout keep the other builds
    if (p.tail_sub || p.m_indices || p.launch_tiles > 0 || p.splitk > 1 || p.kb_n < 2) return DGA_E_TILING;
    const bool nt = p.b_nt != 0;
#ifdef DGA_BXG_KNOBS      // development builds: $DGA_BXG_KNOB & 2 = the build without the stagger (A/B runs in one process)
    if (const char *e = std::getenv("DGA_BXG_KNOB"); e && (std::atoi(e) & 2)) {
        if (p.k % 128) return nt ? launch_bf16x_grouped_one<true, true, false>(p, stream) : launch_bf16x_grouped_one<true, false, false>(p, stream);
        return nt ? launch_bf16x_grouped_one<false, true, false>(p, stream) : launch_bf16x_grouped_one<false, false, false>(p, stream);
    }
#endif
    if (p.row_index) {     // indexed rows: the build that finds every row, its scales and its result row through the slot table
        if (p.k % 128) return nt ? launch_bf16x_grouped_one<true, true, true, true>(p, stream) : launch_bf16x_grouped_one<true, false, true, true>(p, stream);
        return nt ? launch_bf16x_grouped_one<false, true, true, true>(p, stream) : launch_bf16x_grouped_one<false, false, true, true>(p, stream);
    }
    if (p.k % 128) return nt ? launch_bf16x_grouped_one<true, true, true>(p, stream) : launch_bf16x_grouped_one<true, false, true>(p, stream);
    return nt ? launch_bf16x_grouped_one<false, true, true>(p, stream) : launch_bf16x_grouped_one<false, false, true>(p, stream);
}
}
