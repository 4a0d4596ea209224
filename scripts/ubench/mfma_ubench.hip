// Micro-benchmark: what the matrix pipe sustains for v_mfma_f32_16x16x128_f8f6f4 alone and with the
// block-scale promotion FMAs beside it (development aid, not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>  // 0: MFMA only (accumulate chain x4 indep), 1: MFMA(C=0) + 4 FMA pipelined lag 3, 2: 32x32x64 variant
__global__ void __launch_bounds__(512) k(const int *seed, float *out, int iters)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = seed[(threadIdx.x * 8 + i) & 4095]; b[i] = seed[(threadIdx.x * 8 + i + 2048) & 4095]; }
    v4f acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
    float s = 1.0001f;
    v2f acc2[32];
    for (int i = 0; i < 32; ++i) acc2[i] = v2f{0, 0};
    if (MODE == 2) {   // 32x32x64 form, MFMA only, 4 independent accumulators (same flops per wave-iteration as 16 x 16x16x128)
        v16f a32[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) a32[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    a32[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, a32[i], 0, 0, 0, 0, 0, 0);
        }
        float r = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += a32[i][j];
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;
        return;
    }
    if (MODE == 4) {   // 32x32x64: two chained MFMAs per 128-wide scale block (C = 0, then C = part) + 16 promotion FMAs, lag 1 block
        v16f acc32[4], part32[2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc32[i][j] = 0.f;
        for (int j = 0; j < 16; ++j) { part32[0][j] = 0.f; part32[1][j] = 0.f; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4 + 1; ++i) {
                if (i < 4) {
                    asm volatile("" : "+v"(a));
                    v16f z16; for (int j = 0; j < 16; ++j) z16[j] = 0.f;
                    v16f t = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, z16, 0, 0, 0, 0, 0, 0);
                    part32[i & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, t, 0, 0, 0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (i >= 1) {
                    const int j = i - 1;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc32[j][q] = __builtin_fmaf(part32[j & 1][q], s, acc32[j][q]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float r = 0;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc32[i][j];
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;
        return;
    }
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
        }
    } else {
        v4f part[4];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16 + 3; ++i) {
                if (i < 16) {
                    asm volatile("" : "+v"(a));  // opaque: no CSE of the loop-invariant MFMA
                    part[i & 3] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, v4f{0, 0, 0, 0}, 0, 0, 0, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (MODE == 3 && i >= 3) {   // two v_pk_fma_f32 instead of four v_fma_f32 (accumulators kept as register pairs)
                    const int j = i - 3;
                    const v2f sv = v2f{s, s};
                    const v2f plo = __builtin_shufflevector(part[j & 3], part[j & 3], 0, 1);
                    const v2f phi = __builtin_shufflevector(part[j & 3], part[j & 3], 2, 3);
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[2 * j]) : "v"(plo), "v"(sv));
                    asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc2[2 * j + 1]) : "v"(phi), "v"(sv));
                } else if (i >= 3) {
                    const int j = i - 3;
                    acc[j].x = __builtin_fmaf(part[j & 3].x, s, acc[j].x);
                    acc[j].y = __builtin_fmaf(part[j & 3].y, s, acc[j].y);
                    acc[j].z = __builtin_fmaf(part[j & 3].z, s, acc[j].z);
                    acc[j].w = __builtin_fmaf(part[j & 3].w, s, acc[j].w);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    if (MODE == 3) for (int i = 0; i < 32; ++i) r += acc2[i].x + acc2[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

int main(int argc, char **argv)
{
    int iters = 2000;
    int *seed; float *out;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 8 * 512 * 4);
    std::vector<int> h(4096);
    const int randomise = argc > 1 ? atoi(argv[1]) : 1;
    srand(7);
    for (int i = 0; i < 4096; ++i) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) {
            unsigned byte = randomise ? (rand() & 0xFF) : 0x38;
            if ((byte & 0x7F) == 0x7F) byte &= 0x80;      // no NaN
            if ((byte & 0x78) > 0x58) byte &= 0xDF;        // keep magnitudes moderate (no overflow to inf in acc chains)
            v |= byte << (8 * b);
        }
        h[i] = (int)v;
    }
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("operands: %s\n", randomise ? "random e4m3 bytes" : "constant 0x38");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 5; ++mode)
        for (int threads : {256, 512}) {
            const int reps = argc > 2 ? atoi(argv[2]) : 3;   // argv[2]: launches per mode; the LAST one is reported (300+ = sustained clocks)
            for (int rep = 0; rep < reps; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                else hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                double flops = 2.0 * 16 * 16 * 128 * 16.0 * iters * (threads / 64) * 256;
                if (rep == reps - 1) printf("mode %d (%s) waves/SIMD %d: %.3f ms  %.0f TFLOP/s\n", mode, mode == 1 ? "MFMA+4FMA lag3" : mode == 2 ? "32x32x64 MFMA only" : mode == 3 ? "MFMA+2 pk_fma lag3" : mode == 4 ? "32x32x64 x2 + 16 FMA" : "MFMA only", threads / 256, ms, flops / ms / 1e9);
            }
        }
    return 0;
}
