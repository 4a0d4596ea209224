// fp8 kernel menu, part G: the one-launch workgroup split-K builds for short-M problems (gemm_fp8_wsk_kernel.hpp,
// kernelSerial DGA_KERNEL_SPLITK_WORKGROUP): 8 waves = 8 K slices of one output tile, partial tiles combined in LDS -- operands
// staged through per-wave LDS-DMA rings (M <= 32; pass-by-pass and continuous-ring builds) or streamed global -> registers (M <= 64).
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_wsk_kernel.hpp"
namespace dga {

template <int TM, int TNMAX, int D, bool KTAIL>
static int launch_wsk_one(const GemmParams &p, unsigned grid, hipStream_t stream)
{
    auto kfn = gemm_fp8_wsk_kernel<TM, TNMAX, D, KTAIL>;
    constexpr int kLds = 8 * TM * 16 * TNMAX * 16 * 4;
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kLds, stream, p);
    return record_hip(hipGetLastError());
}

template <int TM, int TN, int D, bool KTAIL, int WAVES = 8, int MATH = 0>
static int launch_wskd_one(const GemmParams &p, unsigned grid, hipStream_t stream)
{
    auto kfn = gemm_fp8_wskd_kernel<TM, TN, D, KTAIL, WAVES, MATH>;
    constexpr int kLds = WAVES * D * ((TM + TN) * 16 * 128 + (TM * 16 + 2 + 63) / 64 * 256);
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(WAVES * 64), kLds, stream, p);
    return record_hip(hipGetLastError());
}

template <int TM, int TN, int D, bool KTAIL, int MATH>
static int launch_wskc_one(const GemmParams &p, unsigned grid, hipStream_t stream)
{
    auto kfn = gemm_fp8_wskc_kernel<TM, TN, D, KTAIL, MATH>;
    constexpr int kLds = 8 * (D * ((TM + TN) * 16 * 128 + (TM * 16 + 2 + 63) / 64 * 256) + TM * 16 * TN * 16 * 4);
    static_assert(kLds <= 160 * 1024, "LDS of one CU");
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), kLds, stream, p);
    return record_hip(hipGetLastError());
}

// the LDS-DMA staged builds (M <= 32): one workgroup per CU (or per n-tile where there are fewer), each walking its n-tiles TN at a time.
// (4-wave forms for up to 64 rows -- four K slices, twice the ring per wave -- were built and measured too: within 5 % of the tile
//  kernels on 7 of 72 cold shapes, 10-100 % behind elsewhere, profiles/r04_sweep_wskd/table_m64.txt; they are not in the menu.  The
//  kernel keeps its WAVES parameter.)
template <int MATH>
static int launch_wsk_dma_math(const GemmParams &p, hipStream_t stream)
{
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.splitk > 1 || p.tail_sub || p.m > 32 || p.m <= 0 || (p.k % 16) ||
        p.k <= 0 || (reinterpret_cast<uintptr_t>(p.a) & 15) || (reinterpret_cast<uintptr_t>(p.b) & 15) || (p.lda & 15) || (p.ldb & 15) ||
        static_cast<int64_t>(p.m) * p.lda >= 0x7FFFFFFFll ||
        // B is addressed through a descriptor rebased at every pass's first row: 32-bit offsets cover the pass's rows (at most 96) as
        // long as 96 rows fit 2 GB -- whatever N is (a weight matrix of more than 2 GB is fine, a single row of more than 22 MB is not)
        static_cast<int64_t>(p.ldb) * 96 >= 0x7FFFFFFFll)
        return DGA_E_TILING;
    const int nt = (p.n + 15) / 16;
    const int64_t cus = device_cus();
    const unsigned g = static_cast<unsigned>(nt < cus ? nt : cus);
    const bool kt = (p.k % 128) != 0;
    // (fewer workgroups with three n-tiles each -- a third less A traffic per B byte -- measured worse: 8 x 7168 x 18432 27.9 -> 34.9 us;
    //  every CU streaming counts for more than the A bytes)
    if (p.m > 16) return kt ? launch_wskd_one<2, 2, 2, true, 8, MATH>(p, g, stream) : launch_wskd_one<2, 2, 2, false, 8, MATH>(p, g, stream);
    // n-tiles per workgroup -> the build that walks them in the fewest passes (every pass re-streams the A rows and pays a round
    // trip); at equal passes the narrower one (deeper ring)
    const int per = static_cast<int>((nt + g - 1) / g);
    // more n-tiles per workgroup than one pass holds: the continuous-ring builds (gemm_fp8_wskc_kernel: the refill runs on across
    // the passes).  Three tiles per pass where that saves a pass without leaving a single tile to the last one, two otherwise
    // (cold, M = 8: 18432 x 7168 30.8 -> 28.7 us, 16384 x 7168 26.6 -> 24.7, 28672 x 4096 31.3 -> 27.9; $DGA_WSK_CONT = 0 keeps the
    // pass-by-pass build).
    static const int cont_env = [] { const char *e = std::getenv("DGA_WSK_CONT"); return e ? std::atoi(e) : 1; }();
    const int p2 = (per + 1) / 2, p3 = (per + 2) / 3;
    if (cont_env && per >= 4) {
        if (p3 < p2 && per % 3 != 1)
            return kt ? launch_wskc_one<1, 3, 2, true, MATH>(p, g, stream) : launch_wskc_one<1, 3, 2, false, MATH>(p, g, stream);
        return kt ? launch_wskc_one<1, 2, 2, true, MATH>(p, g, stream) : launch_wskc_one<1, 2, 2, false, MATH>(p, g, stream);
    }
    if (per <= 1) return kt ? launch_wskd_one<1, 1, 4, true, 8, MATH>(p, g, stream) : launch_wskd_one<1, 1, 4, false, 8, MATH>(p, g, stream);
    if ((per + 2) / 3 < (per + 1) / 2) return kt ? launch_wskd_one<1, 3, 2, true, 8, MATH>(p, g, stream) : launch_wskd_one<1, 3, 2, false, 8, MATH>(p, g, stream);
    return kt ? launch_wskd_one<1, 2, 3, true, 8, MATH>(p, g, stream) : launch_wskd_one<1, 2, 3, false, 8, MATH>(p, g, stream);
}

int launch_wsk_dma(const GemmParams &p, hipStream_t stream, int math)
{
    return math ? launch_wsk_dma_math<1>(p, stream) : launch_wsk_dma_math<0>(p, stream);
}

// rows of the build that takes M rows (0: none) and the n-tiles (16 columns) one of its workgroups may own
int wsk_rows(int m) { return m <= 16 ? 16 : (m <= 32 ? 32 : (m <= 64 ? 64 : 0)); }
int wsk_max_ntiles(int m) { return m <= 32 ? 5 : 4; }

int launch_wsk(const GemmParams &p, hipStream_t stream)
{
    // dense problems of at most 64 rows, 16-byte K chunks; everything else keeps the tile kernels
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.splitk > 1 || p.tail_sub || p.m > 64 || p.m <= 0 || (p.k % 16) ||
        p.k <= 0 || static_cast<int64_t>(p.m) * p.lda >= 0x7FFFFFFFll || static_cast<int64_t>(p.ldb) * 96 >= 0x7FFFFFFFll)
        return DGA_E_TILING;
    const int nt = (p.n + 15) / 16, tnmax = wsk_max_ntiles(p.m);
    const int64_t cus = device_cus();
    // one workgroup per CU where the columns allow it (each owns nt / G n-tiles, balanced to within one), more where a
    // workgroup would otherwise own more n-tiles than its registers hold
    int64_t grid = nt < cus ? nt : cus;
    if ((nt + grid - 1) / grid > tnmax) grid = (nt + tnmax - 1) / tnmax;
    const bool kt = (p.k % 128) != 0;
    const unsigned g = static_cast<unsigned>(grid);
    if (p.m <= 16) return kt ? launch_wsk_one<1, 5, 3, true>(p, g, stream) : launch_wsk_one<1, 5, 3, false>(p, g, stream);
    if (p.m <= 32) return kt ? launch_wsk_one<2, 5, 2, true>(p, g, stream) : launch_wsk_one<2, 5, 2, false>(p, g, stream);
    return kt ? launch_wsk_one<4, 4, 2, true>(p, g, stream) : launch_wsk_one<4, 4, 2, false>(p, g, stream);
}
}
