// Definitions behind dga_fp8_menu.hpp; included only by the dga_launch_menu_*.hip units, which instantiate their share.
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>

#include "dga_fp8_menu.hpp"

namespace dga {

template <class Cfg, int PP, bool KTAIL, bool CLK>
static int launch_one(const GemmParams &p, hipStream_t stream)
{
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, PP, KTAIL, CLK>;
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
    if (p.m_indices && Cfg::kBM > DGA_CONTIGUOUS_M_ALIGNMENT) grid *= 2;  // pass-1 copies for straddling tiles
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

template <class Cfg, int PP, bool CLK>
int launch_cfg(const GemmParams &p, hipStream_t stream)
{
    if constexpr (CLK) {
        if (p.k % 128) return DGA_E_TILING;
        return launch_one<Cfg, PP, false, true>(p, stream);
    } else {
        return (p.k % 128) ? launch_one<Cfg, PP, true, false>(p, stream) : launch_one<Cfg, PP, false, false>(p, stream);
    }
}

#define DGA_MENU_INSTANTIATE(BM, BN, WM, WN, ST, PP) \
    template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST>, PP, false>(const GemmParams &, hipStream_t);
#define DGA_MENU_INSTANTIATE_LC(BM, BN, WM, WN, ST, PP) \
    template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST, 4>, PP, false>(const GemmParams &, hipStream_t);
#define DGA_MENU_INSTANTIATE_CLK_LC(BM, BN, WM, WN, ST, PP) \
    template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST, 4>, PP, true>(const GemmParams &, hipStream_t);
#define DGA_MENU_INSTANTIATE_CLK(BM, BN, WM, WN, ST, PP) \
    template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST>, PP, true>(const GemmParams &, hipStream_t);

}  // namespace dga
