// Micro-benchmark for the strict policy's inner step (development aid): v_mfma_f32_16x16x4_f32 chains on 8 accumulators (the 32 x 64
// wave tile: 2 m-tiles x 4 n-tiles), operands in registers, with and without the per-value byte extraction + v_cvt_f32_fp8 that feeds
// them (6 conversions per 8 MFMAs), at 1..4 waves per SIMD.  What does the fp32 matrix pipe sustain in this shape of loop?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int CVT>
__global__ void __launch_bounds__(256) k(const int *seed, float *out, int iters)
{
    int raw[6];
    for (int i = 0; i < 6; ++i) raw[i] = seed[(threadIdx.x * 6 + i) & 4095];
    const int sh = 8 * ((threadIdx.x & 63) >> 4);
    v4f acc[2][4];
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = v4f{0, 0, 0, 0};
    float fa[2] = {1.f, 2.f}, fb[4] = {1.f, 2.f, 3.f, 4.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            if (CVT) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) fa[mt] = __builtin_amdgcn_cvt_f32_fp8((int)((unsigned)raw[mt] >> sh), 0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) fb[nt] = __builtin_amdgcn_cvt_f32_fp8((int)((unsigned)raw[2 + nt] >> sh), 0);
#pragma unroll
                for (int i = 0; i < 6; ++i) raw[i] = raw[i] * 5 + d;     // the next step's bytes (keeps the conversions live)
            } else {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) asm volatile("" : "+v"(fa[mt]));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) asm volatile("" : "+v"(fb[nt]));
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[nt], fa[mt], acc[mt][nt], 0, 0, 0);
        }
    }
    float r = 0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 4; ++j) r += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main()
{
    int *seed; float *out;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 256 * 16 * 4);
    int h[4096]; for (int i = 0; i < 4096; ++i) h[i] = rand();
    hipMemcpy(seed, h, sizeof(h), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int cvt = 0; cvt < 2; ++cvt)
        for (int wps = 1; wps <= 4; ++wps) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                for (int l = 0; l < 20; ++l) {
                    if (cvt) hipLaunchKernelGGL(k<1>, dim3(256 * wps), dim3(256), 0, 0, seed, out, iters);
                    else hipLaunchKernelGGL(k<0>, dim3(256 * wps), dim3(256), 0, 0, seed, out, iters);
                }
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            const double flops = 2.0 * 1024 * 32.0 * iters * 4 * 256 * wps * 20;   // 32 MFMAs per iteration per wave, 4 waves per workgroup
            printf("%s  waves/SIMD %d: %8.3f ms  %7.1f TFLOP/s  (%.3f of 157.3)\n", cvt ? "MFMA + byte extraction + v_cvt_f32_fp8" : "MFMA only                             ",
                   wps, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3);
        }
    return 0;
}
