// Diagnostics of the C ABI: the clock the chip holds inside the dense kernel's main loop.
// SURVEY.md 8(d) asks for the measured clock beside the vendor peak (peak_fp8 = CUs x clk x 8192): an MFMA-dense loop
// runs well under the 2.4 GHz the 5 PFLOP/s figure assumes (MI355X_MICROARCH.md "DVFS give-back").  The probe launches
// the CLK = true instantiation of the very kernel the tiling selects -- same code plus one s_memtime / s_memrealtime
// pair in front of and behind the k loop -- and reports median(shader ticks / 100 MHz ticks) over the waves.
// The product kernels carry no stamp; this entry point is never on the hot path (it synchronises the stream).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <vector>

#include "dga_fp8_menu_impl.hpp"

namespace dga {
DGA_MENU_CLK(DGA_MENU_INSTANTIATE_CLK)
DGA_MENU_CLK_LC(DGA_MENU_INSTANTIATE_CLK_LC)
}

extern "C" int dga_gemm_fp8_loop_clock(const void *a, const float *sfa, const void *b, const float *sfb, void *out, int m,
                                       int n, int k, const dga_tiling_t *tiling, void *scratch, size_t scratch_bytes,
                                       int launches, void *stream, float *clock_mhz, float *loop_us)
{
    if (!tiling || !scratch || !clock_mhz) return DGA_E_NULL;
    if (!tiling->m1 || !tiling->n1 || launches < 1) return DGA_E_RANGE;
    const size_t tiles = static_cast<size_t>((m + tiling->m1 - 1) / tiling->m1) * ((n + tiling->n1 - 1) / tiling->n1);
    const size_t waves = tiles * 8;                       // both clock builds run 8 waves per workgroup
    const size_t need = waves * 2 * sizeof(unsigned long long);
    if (scratch_bytes < need) return DGA_E_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (int rc = dga::record_hip(hipMemsetAsync(scratch, 0, need, s))) return rc;
    for (int i = 0; i < launches; ++i) {                  // the last launch's stamps are the ones read back
        int rc = dga::run_fp8(a, sfa, b, sfb, out, nullptr, nullptr, 1, 1, m, n, k, 0, tiling, nullptr, 0, s,
                              static_cast<unsigned long long *>(scratch), nullptr);
        if (rc != DGA_OK) return rc;
    }
    std::vector<unsigned long long> h(waves * 2);
    if (int rc = dga::record_hip(hipMemcpyAsync(h.data(), scratch, need, hipMemcpyDeviceToHost, s))) return rc;
    if (int rc = dga::record_hip(hipStreamSynchronize(s))) return rc;
    std::vector<double> mhz, us;
    for (size_t w = 0; w < waves; ++w)
        if (h[2 * w + 1]) {
            mhz.push_back(100.0 * static_cast<double>(h[2 * w]) / static_cast<double>(h[2 * w + 1]));
            us.push_back(static_cast<double>(h[2 * w + 1]) / 100.0);
        }
    if (mhz.empty()) return DGA_E_TILING;
    std::nth_element(mhz.begin(), mhz.begin() + mhz.size() / 2, mhz.end());
    std::nth_element(us.begin(), us.begin() + us.size() / 2, us.end());
    *clock_mhz = static_cast<float>(mhz[mhz.size() / 2]);
    if (loop_us) *loop_us = static_cast<float>(us[us.size() / 2]);
    return DGA_OK;
}
