// fp8 tile-kernel menu, part K: the one-launch Stream-K build of the 256 x 256 continuous kernel (gemm_fp8_streamk_kernel.hpp,
// kernelSerial DGA_KERNEL_STREAMK_ONE_LAUNCH): the reference's kernel type 4 read for CDNA4 -- the raster's k blocks cut evenly over
// the CUs, fp32 partial tiles through the caller's workspace, reduced in k order by the workgroup that holds a tile's first blocks.
#include <atomic>
#include <cstdlib>
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_streamk_kernel.hpp"
namespace dga {

size_t streamk_workspace_bytes() { return static_cast<size_t>(device_cus()) * (256 * 256 * 4 + 8) + 256; }

template <int MATH>
static int launch_streamk_one(const GemmParams &p, const StreamKArgs &sk, unsigned grid, hipStream_t stream)
{
    typedef GemmCfg<256, 256, 4, 2, 2> Cfg;
    auto kfn = gemm_fp8_blockscaled_nt_streamk_kernel<Cfg, MATH>;
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
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p, sk);
    return record_hip(hipGetLastError());
}

// ws: the caller's workspace, at least streamk_workspace_bytes().  DGA_E_TILING: not a problem this kernel takes (the caller then
// runs the tiling's tile kernel).  ue8m0: the hardware-scale form (power-of-two block scales).
int launch_streamk(const GemmParams &p, void *ws, size_t ws_bytes, bool ue8m0, hipStream_t stream)
{
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.splitk > 1 || p.tail_sub || p.stamps || p.launch_tiles) return DGA_E_TILING;
    if ((p.m % 256) || (p.n % 256) || (p.k % 128) || p.kb_n < 2) return DGA_E_TILING;
    // one workgroup per CU (they wait for one another's partial tiles: all of them must be resident at once -- where a CU mask narrows
    // the queue that cannot be promised, and the caller runs the tile kernel: dga_launch_menu_m.hip coresident_workgroups)
    const int64_t grid = coresident_workgroups(stream);
    if (grid <= 0) return DGA_E_TILING;
    // (a workspace that is missing, short or misaligned: the tile kernel too -- the documented fall-back, not an error)
    if (!ws || ws_bytes < static_cast<size_t>(grid) * (256 * 256 * 4) + static_cast<size_t>(grid) * 8) return DGA_E_TILING;
    if (reinterpret_cast<uintptr_t>(ws) & 15) return DGA_E_TILING;
    StreamKArgs sk;
    sk.partials = static_cast<float *>(ws);
    sk.flags = reinterpret_cast<unsigned long long *>(sk.partials + static_cast<size_t>(grid) * (256 * 256));
    // A flag is raised when it holds this launch's epoch: 64 mixed bits no earlier launch used and stale workspace bytes will not
    // hold -- nothing to zero.  A launch that is being CAPTURED into a graph is replayed with the same arguments, so there the flags
    // are zeroed by a memset node in front of the kernel and the epoch is a constant.
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cap) != hipSuccess) cap = hipStreamCaptureStatusNone;
    if (cap != hipStreamCaptureStatusNone) {
        if (int rc = record_hip(hipMemsetAsync(sk.flags, 0, static_cast<size_t>(grid) * 8, stream))) return rc;
        sk.epoch = 1ull;
    } else {
        static std::atomic<unsigned long long> launches{0};
        const unsigned long long e = launches.fetch_add(1) + 1;
        sk.epoch = (e * 0x9E3779B97F4A7C15ull) | 1ull;     // odd, never 0 (and never 1: e >= 1 gives at least 2^63-ish mixed bits)
        if (sk.epoch == 1ull) sk.epoch = 3ull;
    }
    return ue8m0 ? launch_streamk_one<2>(p, sk, static_cast<unsigned>(grid), stream)
                 : launch_streamk_one<0>(p, sk, static_cast<unsigned>(grid), stream);
}

}  // namespace dga
