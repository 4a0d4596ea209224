// fp8 tile-kernel menu, part D: the persistent loader-wave builds (dga_fp8_menu.hpp, gemm_fp8_persistent_kernel.hpp).
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_persistent_kernel.hpp"
#include "gemm_fp8_cont_persistent_kernel.hpp"
namespace dga {

template <class Cfg, bool KTAIL>
static int launch_persistent_one(const GemmParams &p, hipStream_t stream)
{
    auto kfn = gemm_fp8_blockscaled_nt_persistent_kernel<Cfg, KTAIL>;
    // the stage ring, then the loader waves' row-table slots (indexed form): one dword per lane and DMA piece
    constexpr int kLds = Cfg::LDS_BYTES + Cfg::DMA_WAVES * (Cfg::A_ITERS + Cfg::SC_ITERS) * 256;
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    // one workgroup per CU (a stage ring of this size leaves room for one), fewer when the raster is smaller
    const int64_t tiles = static_cast<int64_t>(p.groups) * p.tiles_m * p.tiles_n;
    const int64_t cus = device_cus();
    const unsigned grid = static_cast<unsigned>(tiles < cus ? tiles : cus);
    if (grid == 0) return DGA_OK;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), kLds, stream, p);
    return record_hip(hipGetLastError());
}

template <class Cfg>
int launch_persistent(const GemmParams &p, hipStream_t stream)
{
    // the persistent builds take whole rasters only: no split-K slabs, no quarter-tile tail, tiles no taller than the
    // contiguous layout's segment alignment
    if (p.splitk > 1 || p.tail_sub || p.launch_tiles > 0) return DGA_E_TILING;
    if (p.m_indices && Cfg::kBM > DGA_CONTIGUOUS_M_ALIGNMENT) return DGA_E_TILING;
    return (p.k % 128) ? launch_persistent_one<Cfg, true>(p, stream) : launch_persistent_one<Cfg, false>(p, stream);
}

// The loader-wave build of the 128 x 256 tile whose loaders take rows that start at any byte (K % 16 != 0 without a padded copy;
// gemm_fp8_kernel.hpp UNAL): dense problems only.
int launch_unaligned(const GemmParams &p, hipStream_t stream)
{
    typedef GemmCfg<128, 256, 2, 2, 3, 4> Cfg;
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.tail_sub || p.stamps) return DGA_E_TILING;
    // 32-bit byte offsets from the matrices' first bytes
    if (static_cast<int64_t>(p.m) * p.lda >= 0x7FFFFFFFll || static_cast<int64_t>(p.n) * p.ldb >= 0x7FFFFFFFll) return DGA_E_TILING;
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, 0, true, false, 0, true>;
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    const unsigned grid = p.launch_tiles > 0 ? static_cast<unsigned>(p.launch_tiles) : static_cast<unsigned>(p.groups) * p.tiles_m * p.tiles_n;
    if (grid == 0) return DGA_OK;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

int launch_cont_persistent(const GemmParams &p, hipStream_t stream)
{
    typedef GemmCfg<256, 256, 4, 2, 2> Cfg;
    // dense rasters of full tiles, at least two k blocks (the refill slots look one tile ahead)
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.splitk > 1 || p.tail_sub || p.stamps) return DGA_E_TILING;
    if ((p.m % 256) || (p.n % 256) || (p.k % 128) || p.kb_n < 2) return DGA_E_TILING;
    auto kfn = gemm_fp8_blockscaled_nt_cont_persistent_kernel<Cfg>;
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
    const int64_t tiles = p.launch_tiles > 0 ? p.launch_tiles : static_cast<int64_t>(p.tiles_m) * p.tiles_n;
    const int64_t cus = device_cus();
    const unsigned grid = static_cast<unsigned>(tiles < cus ? tiles : cus);
    if (grid == 0) return DGA_OK;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, stream, p);
    return record_hip(hipGetLastError());
}

#define DGA_MENU_INSTANTIATE_PS(BM, BN, WM, WN, ST, PP) \
    template int launch_persistent<GemmCfg<BM, BN, WM, WN, ST, 4>>(const GemmParams &, hipStream_t);
DGA_MENU_LC(DGA_MENU_INSTANTIATE_PS)
}
