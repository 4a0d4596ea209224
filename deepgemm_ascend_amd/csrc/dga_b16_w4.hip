// The four-wave 32x32x16 build of the 16-bit 256 x 256 tile (gemm_b16_w4_kernel.hpp).  Its own translation unit: compiled WITHOUT
// -amdgpu-mfma-vgpr-form (Makefile NOFORM_dga_b16_w4), so that a wave's 256 accumulators live in AGPRs.
#include <hip/hip_runtime.h>
#include <mutex>

#include "dga_hip.h"
#include "dga_internal.hpp"
#define DGA_B16_TILE_KERNEL_ONLY
#include "gemm_b16_w4_kernel.hpp"

namespace dga {

template <bool BF16>
static int launch_w4_one(const B16Params &p, hipStream_t stream)
{
    using Cfg = GemmCfg<256, 256, 2, 2, 2>;
    auto kfn = gemm_b16_w4_kernel<BF16>;
    constexpr int lds = Cfg::STAGES * (Cfg::A_BYTES + Cfg::B_BYTES);
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    });
    if (record_hip(attr_err[dev]) != DGA_OK) return DGA_E_HIP;
    hipLaunchKernelGGL(kfn, dim3(static_cast<unsigned>(p.tiles_m) * p.tiles_n), dim3(256), lds, stream, p);
    return record_hip(hipGetLastError());
}

// operator form only: batch 1, no split-K, no tail, p.z16 set, p.k % 64 == 0, 16-byte aligned operands with 16-byte aligned rows
int launch_b16_w4(const B16Params &p, bool bf16, hipStream_t stream)
{
    if (p.batch != 1 || p.splitk > 1 || p.tail_sub || p.launch_tiles || !p.z16 || (p.k % 64)) return DGA_E_TILING;
    return bf16 ? launch_w4_one<true>(p, stream) : launch_w4_one<false>(p, stream);
}

}  // namespace dga
