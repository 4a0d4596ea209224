// fp8 tile-kernel menu, part I: the hardware-scale builds (gemm_fp8_kernel.hpp MATH = 2, DGA_POLICY_UE8M0_SCALES) -- the block
// scales ride in the E8M0 operands of v_mfma_scale_f32_16x16x128_f8f6f4 and the MFMA accumulates in place, for scale tensors
// whose values are exact powers of two.  The continuous 256x256 tile and the three-stage tiles with and without loader waves.
#include "dga_fp8_menu_impl.hpp"
namespace dga {

template <class Cfg, int PP, bool KTAIL>
static int launch_ue8m0_one(const GemmParams &p, hipStream_t stream)
{
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, PP, KTAIL, false, 2>;
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
    unsigned grid = p.launch_tiles > 0 ? static_cast<unsigned>(p.launch_tiles)
                                       : static_cast<unsigned>(p.groups) * p.tiles_m * p.tiles_n;
    if (grid == 0) return DGA_OK;
    if (p.m_indices && Cfg::kBM > DGA_CONTIGUOUS_M_ALIGNMENT) grid *= 2;  // pass-1 copies for straddling tiles
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

template <class Cfg, int PP>
static int launch_ue8m0_cfg(const GemmParams &p, hipStream_t stream)
{
    return (p.k % 128) ? launch_ue8m0_one<Cfg, PP, true>(p, stream) : launch_ue8m0_one<Cfg, PP, false>(p, stream);
}

// (bm, bn) = the tile; loaders: the tiling asks for the loader-wave build; cont: the continuous pipeline (256 x 256 only).
// DGA_E_TILING: no hardware-scale build of that tile -- the caller runs the promotion build (same outputs up to fp32 rounding order).
int launch_ue8m0(int bm, int bn, bool loaders, bool cont, const GemmParams &p, hipStream_t stream)
{
    if (p.stamps) return DGA_E_TILING;
    if (bm == 256 && bn == 256) return cont ? launch_ue8m0_cfg<GemmCfg<256, 256, 4, 2, 2>, 2>(p, stream)
                                            : launch_ue8m0_cfg<GemmCfg<256, 256, 4, 2, 2>, 0>(p, stream);
    if (bm == 128 && bn == 256) return loaders ? launch_ue8m0_cfg<GemmCfg<128, 256, 2, 2, 3, 4>, 0>(p, stream)
                                               : launch_ue8m0_cfg<GemmCfg<128, 256, 2, 2, 3>, 0>(p, stream);
    if (bm == 128 && bn == 128) return loaders ? launch_ue8m0_cfg<GemmCfg<128, 128, 2, 2, 3, 4>, 0>(p, stream)
                                               : launch_ue8m0_cfg<GemmCfg<128, 128, 2, 2, 3>, 0>(p, stream);
    if (bm == 64 && bn == 256) return loaders ? launch_ue8m0_cfg<GemmCfg<64, 256, 1, 4, 3, 4>, 0>(p, stream)
                                              : launch_ue8m0_cfg<GemmCfg<64, 256, 1, 4, 3>, 0>(p, stream);
    if (bm == 64 && bn == 128) return loaders ? launch_ue8m0_cfg<GemmCfg<64, 128, 1, 4, 3, 4>, 0>(p, stream)
                                              : launch_ue8m0_cfg<GemmCfg<64, 128, 1, 4, 3>, 0>(p, stream);
    return DGA_E_TILING;
}

}  // namespace dga
