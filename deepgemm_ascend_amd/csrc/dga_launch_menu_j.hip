// fp8 tile-kernel menu, part J: the four-wave hardware-scale build of the 256 x 256 tile (gemm_fp8_kernel.hpp MATH = 2 on
// GemmCfg<256, 256, 2, 2>): one wave per SIMD, wave tile 128 x 128 = 256 fp32 accumulators per lane.  Possible only because the
// MFMA accumulates in place -- no vector instruction ever touches an accumulator, so they may live in AGPRs (this unit is compiled
// WITHOUT -amdgpu-mfma-vgpr-form: Makefile NOFORM_dga_launch_menu_j) -- and a wave's fragment reads per k block fall from
// 24 KB x 8 waves to 32 KB x 4 waves: a third fewer LDS bytes per flop in a loop that runs at constant power.
#include "dga_fp8_menu_impl.hpp"
namespace dga {

template <bool KTAIL>
static int launch_ue8m0_w4_one(const GemmParams &p, hipStream_t stream)
{
    typedef GemmCfg<256, 256, 2, 2, 2> Cfg;
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, 2, KTAIL, false, 2>;
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
    if (p.m_indices) grid *= 2;  // pass-1 copies for tiles that straddle two groups (256 rows > the segment alignment)
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

int launch_ue8m0_w4(const GemmParams &p, hipStream_t stream)
{
    if (p.stamps) return DGA_E_TILING;
    return (p.k % 128) ? launch_ue8m0_w4_one<true>(p, stream) : launch_ue8m0_w4_one<false>(p, stream);
}

// ---- bf16-exact arithmetic for power-of-two scales (gemm_fp8_kernel.hpp MATH = 3, DGA_POLICY_BF16_EXACT | DGA_POLICY_UE8M0_SCALES): the
//      scales are folded into the A conversions and the bf16 MFMA chain accumulates in place, so the accumulators sit in AGPRs here too:
//      the 128 x 256 tile on FOUR waves (wave tile 64 x 128, 1.5 conversions per MFMA and no promotion).  Every layout the tile
//      kernel takes (dense, masked, contiguous, indexed, split-K).
template <class Cfg, bool KTAIL>
static int launch_bf16u_one(const GemmParams &p, hipStream_t stream)
{
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, 0, KTAIL, false, 3>;
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

int launch_bf16u(int bm, int bn, const GemmParams &p, hipStream_t stream)
{
    if (p.stamps || p.tail_sub) return DGA_E_TILING;
    if (bm == 128 && bn == 256) {
        typedef GemmCfg<128, 256, 2, 2, 3> Cfg;
        return (p.k % 128) ? launch_bf16u_one<Cfg, true>(p, stream) : launch_bf16u_one<Cfg, false>(p, stream);
    }
    if (bm == 64 && bn == 256) {
        typedef GemmCfg<64, 256, 1, 4, 3> Cfg;
        return (p.k % 128) ? launch_bf16u_one<Cfg, true>(p, stream) : launch_bf16u_one<Cfg, false>(p, stream);
    }
    return DGA_E_TILING;
}

}  // namespace dga
