// fp8 tile-kernel menu, part H: the persistent form of the bf16-exact policy's 128 x 256 in-register build
// (gemm_fp8_bf16x_persistent_kernel.hpp, dispatchPolicyTag 7): one workgroup per CU walks the raster, the LDS ring runs across tiles.
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_bf16x_persistent_kernel.hpp"
namespace dga {

template <bool KTAIL>
static int launch_bf16x_persistent_one(const GemmParams &p, hipStream_t stream)
{
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    auto kfn = gemm_fp8_bf16x_persistent_kernel<KTAIL>;
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    const int64_t tiles = p.launch_tiles > 0 ? p.launch_tiles : static_cast<int64_t>(p.groups) * p.tiles_m * p.tiles_n;
    if (tiles == 0) return DGA_OK;
    const unsigned grid = static_cast<unsigned>(std::min<int64_t>(tiles, device_cus()));
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

int launch_bf16x_persistent(const GemmParams &p, hipStream_t stream)
{
    // dense and masked-grouped rasters of at least two k blocks (launch_tiles > 0: the first tiles of a dense raster -- the whole
    // rounds in front of a quarter-tile tail); split-K, the quarter tiles themselves, indexed rows and the contiguous layout keep
    // the one-tile build
    if (p.tail_sub || p.m_indices || p.row_index || (p.launch_tiles > 0 && p.groups != 1) || p.splitk > 1 || p.kb_n < 2) return DGA_E_TILING;
    return (p.k % 128) ? launch_bf16x_persistent_one<true>(p, stream) : launch_bf16x_persistent_one<false>(p, stream);
}
}
