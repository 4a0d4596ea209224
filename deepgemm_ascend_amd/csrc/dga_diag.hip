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


// ---- what the matrix pipe sustains with the operands already in registers ------------------------------------------------
// The ceiling of a DeepSeek-style block-scaled kernel on this device: the kernel's matrix instruction with the fp32
// promotion (and, for the bf16-exact policy, the in-register conversions) beside it, no LDS, no global memory, two waves per
// SIMD on every CU, random e4m3 bytes.  bench.py prices the product kernels against it (`roofline.ceiling_tflops`): clocks
// and the matrix pipe's rate under load differ from box to box by ~10 % (MI355X_MICROARCH.md "DVFS give-back" item 5).
namespace dga {
typedef __bf16 v8bf_t __attribute__((ext_vector_type(8)));

// MODE 0: v_mfma_scale_f32_16x16x128_f8f6f4 (C = 0) + 4 promotion FMAs, pipelined 3 deep (the fast path's inner step)
// MODE 1: 4 chained v_mfma_f32_16x16x32_bf16 + 4 promotion FMAs + 8 conversions per 16x16 tile (the bf16-exact policy's
//         64 x 64 wave tile: 2 conversions per MFMA), every conversion feeding a fragment a LATER MFMA reads
template <int MODE>
__global__ void __launch_bounds__(512) mfma_ceiling_kernel(const int *seed, float *out, int iters)
{
    int raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = seed[(threadIdx.x * 8 + i) & 4095];
    const float s = 1.0001f;
    float r = 0.f;
    if constexpr (MODE == 0) {
        v8i a, b;
        for (int i = 0; i < 8; ++i) { a[i] = raw[i]; b[i] = raw[(i + 3) & 7]; }
        v4f acc[16], part[4];
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 4; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                asm volatile("" : "+v"(a));   // opaque: the loop-invariant MFMA is not hoisted
                part[i & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int j = (i + 13) & 15;   // three steps behind
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[j][q] = __builtin_fmaf(part[j & 3][q], s, acc[j][q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    } else {
        v4i afx[4][4], bfx[2][4];
        auto cvt = [](int w, int half) {
            return half ? __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true))
                        : __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false));
        };
        for (int x = 0; x < 4; ++x)
            for (int q = 0; q < 4; ++q)
                for (int j = 0; j < 4; ++j) {
                    afx[x][q][j] = cvt(raw[(x + q + j) & 7], j & 1);
                    if (x < 2) bfx[x][q][j] = cvt(raw[(x + q + j + 3) & 7], j & 1);
                }
        v4f acc[16], part[4];
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < 4; ++i) part[i] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) {   // 16 tiles = 4 n-tiles x 4 m-tiles, 4 MFMAs each
                const int t = u >> 2, q = u & 3, nt = t >> 2, mt = t & 3, g = u & 15;
                part[t & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf_t, bfx[nt & 1][q]),
                                                                      __builtin_bit_cast(v8bf_t, afx[mt][q]),
                                                                      q == 0 ? v4f{0.f, 0.f, 0.f, 0.f} : part[t & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                // one conversion into the B set the next n-tile reads, one into the A fragment (nt) two tiles after its use
                asm volatile("" : "+v"(raw[g & 7]));
                bfx[(nt + 1) & 1][g >> 2][g & 3] = cvt(raw[g & 7], g & 1);
                afx[(mt + 2) & 3][(nt + q) & 3][g & 3] = cvt(raw[(g + 1) & 7], (g + 1) & 1);
                const int j = (t + 14) & 15;     // two tiles behind
                acc[j][q] = __builtin_fmaf(part[j & 3][q], s, acc[j][q]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
}  // namespace dga

extern "C" int dga_mfma_ceiling(int mode, int launches, void *scratch, size_t scratch_bytes, void *stream, float *tflops)
{
    if (!scratch || !tflops) return DGA_E_NULL;
    if ((mode != 0 && mode != 1) || launches < 1) return DGA_E_RANGE;
    const int cus = static_cast<int>(dga::device_cus());
    const size_t need = 4096 * sizeof(int) + static_cast<size_t>(cus) * 512 * sizeof(float);
    if (scratch_bytes < need) return DGA_E_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int *seed = static_cast<int *>(scratch);
    float *out = reinterpret_cast<float *>(seed + 4096);
    std::vector<int> h(4096);
    unsigned long long x = 88172645463325252ull;   // xorshift: random e4m3 bytes, no NaN codes, moderate magnitudes
    for (auto &w : h) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            unsigned byte = static_cast<unsigned>(x >> 24) & 0xFF;
            if ((byte & 0x7F) == 0x7F) byte &= 0x80;
            if ((byte & 0x78) > 0x58) byte &= 0xDF;
            v |= byte << (8 * b);
        }
        w = static_cast<int>(v);
    }
    if (int rc = dga::record_hip(hipMemcpyAsync(seed, h.data(), 4096 * sizeof(int), hipMemcpyHostToDevice, s))) return rc;
    if (int rc = dga::record_hip(hipStreamSynchronize(s))) return rc;
    const int iters = 400;
    hipEvent_t e0, e1;
    if (int rc = dga::record_hip(hipEventCreate(&e0))) return rc;
    if (int rc = dga::record_hip(hipEventCreate(&e1))) { (void)hipEventDestroy(e0); return rc; }
    float ms = 0.f;
    int rc = DGA_OK;
    for (int i = 0; i < launches && rc == DGA_OK; ++i) {   // the last launch is the one reported (sustained clocks)
        if (i == launches - 1) {
            rc = dga::record_hip(hipEventRecord(e0, s));
            if (rc != DGA_OK) break;   // (the launch's own status below must not overwrite a failed start stamp)
        }
        if (mode == 0) hipLaunchKernelGGL(dga::mfma_ceiling_kernel<0>, dim3(cus), dim3(512), 0, s, seed, out, iters);
        else hipLaunchKernelGGL(dga::mfma_ceiling_kernel<1>, dim3(cus), dim3(512), 0, s, seed, out, iters);
        rc = dga::record_hip(hipGetLastError());
    }
    if (rc == DGA_OK) {
        rc = dga::record_hip(hipEventRecord(e1, s));
        if (rc == DGA_OK) rc = dga::record_hip(hipEventSynchronize(e1));
        if (rc == DGA_OK) rc = dga::record_hip(hipEventElapsedTime(&ms, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc != DGA_OK) return rc;
    // per wave and iteration: 16 tiles of 16 x 16 x 128
    const double flops = 2.0 * 16 * 16 * 128 * 16.0 * iters * 8.0 * cus;
    *tflops = static_cast<float>(flops / (static_cast<double>(ms) * 1e-3) / 1e12);
    return DGA_OK;
}
