// fp8 tile-kernel menu, part E: the bf16-exact builds (dga_fp8_menu.hpp; gemm_fp8_kernel.hpp MATH = 1, dispatchPolicyTag 7).
#include "dga_fp8_menu_impl.hpp"
namespace dga {

template <class Cfg, bool KTAIL>
static int launch_bf16x_one(const GemmParams &p, hipStream_t stream)
{
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, 0, KTAIL, false, 1>;
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
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

template <class Cfg>
int launch_bf16x(const GemmParams &p, hipStream_t stream)
{
    // tiles no taller than the contiguous layout's segment alignment (no second pass)
    if (p.m_indices && Cfg::kBM > DGA_CONTIGUOUS_M_ALIGNMENT) return DGA_E_TILING;
    return (p.k % 128) ? launch_bf16x_one<Cfg, true>(p, stream) : launch_bf16x_one<Cfg, false>(p, stream);
}

#define DGA_MENU_INSTANTIATE_BX(BM, BN, WM, WN, ST, PP) \
    template int launch_bf16x<GemmCfg<BM, BN, WM, WN, ST>>(const GemmParams &, hipStream_t);
DGA_MENU_BX(DGA_MENU_INSTANTIATE_BX)
}
