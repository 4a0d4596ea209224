// fp8 tile-kernel menu, part F: the image build of the bf16-exact policy (gemm_fp8_bf16x_image_kernel.hpp, dispatchPolicyTag 7):
// 128 x 256 tile, both operands converted once per workgroup into a bf16 LDS image; 8 waves (two per SIMD) or 4 (one per SIMD);
// and the A-image build (gemm_fp8_bf16x_aimage_kernel.hpp): only the A-matrix tile goes through the image, 8 waves.
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_bf16x_aimage_kernel.hpp"
namespace dga {

template <class Cfg, bool KTAIL, bool AIMAGE = false>
static int launch_bf16x_image_one(const GemmParams &p, hipStream_t stream)
{
    void (*kfn)(const GemmParams);
    if constexpr (AIMAGE) kfn = gemm_fp8_bf16x_aimage_kernel<KTAIL>;
    else kfn = gemm_fp8_bf16x_image_kernel<Cfg, KTAIL>;
    static_assert(Cfg::LDS_BYTES <= 160 * 1024, "LDS of one CU");
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    const unsigned grid = p.launch_tiles > 0 ? static_cast<unsigned>(p.launch_tiles)
                                             : static_cast<unsigned>(p.groups) * p.tiles_m * p.tiles_n;
    if (grid == 0) return DGA_OK;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

int launch_bf16x_image(const GemmParams &p, int waves, hipStream_t stream)
{
    // dense and masked-grouped rasters (and their split-K form); the contiguous and indexed layouts keep the in-register builds
    if (p.tail_sub || p.m_indices || p.row_index || p.launch_tiles > 0) return DGA_E_TILING;
    if (waves == 1)   // the A-image build
        return (p.k % 128) ? launch_bf16x_image_one<BxAImageCfg, true, true>(p, stream) : launch_bf16x_image_one<BxAImageCfg, false, true>(p, stream);
    if (waves == 4)
        return (p.k % 128) ? launch_bf16x_image_one<BxImageCfg<4>, true>(p, stream) : launch_bf16x_image_one<BxImageCfg<4>, false>(p, stream);
    return (p.k % 128) ? launch_bf16x_image_one<BxImageCfg<8>, true>(p, stream) : launch_bf16x_image_one<BxImageCfg<8>, false>(p, stream);
}
}
