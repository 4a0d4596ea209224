// stamped build of the four-wave 16-bit kernel (csrc/gemm_b16_w4_kernel.hpp with -DDGA_W4_STAMPS): loop cycles, realtime ticks (100 MHz), cycles at vmcnt(0) + barrier.
// build (from deepgemm_ascend_amd/csrc): hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I. -fno-slp-vectorize -x hip ../../scripts/ubench/stamp_b16_w4.hip -o stamp_b16_w4;  run: ./stamp_b16_w4 K launches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define DGA_W4_STAMPS
#define DGA_B16_TILE_KERNEL_ONLY
#include "gemm_b16_w4_kernel.hpp"
using namespace dga;
int main(int argc, char **argv)
{
    const int m = 4096, n = 4096, k = argc > 1 ? atoi(argv[1]) : 4096, reps = argc > 2 ? atoi(argv[2]) : 300;
    uint16_t *x, *y, *z; uint64_t *st;
    hipMalloc(&x, (size_t)m * k * 2); hipMalloc(&y, (size_t)n * k * 2); hipMalloc(&z, (size_t)m * n * 2);
    std::vector<uint16_t> h((size_t)m * k);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (uint16_t)((i * 2654435761u) >> 22 & 0x3ff) - ((i & 1) ? 0 : 0x8000 * ((i >> 3) & 1));
    hipMemcpy(x, h.data(), h.size() * 2, hipMemcpyHostToDevice); hipMemcpy(y, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int tiles = (m / 256) * (n / 256);
    hipMalloc(&st, (size_t)tiles * 4 * 3 * 8);
    B16Params p{};
    p.x = x; p.yt = y; p.z16 = z; p.m = m; p.n = n; p.k = k; p.ldx = k; p.ldy = k; p.batch = 1; p.splitk = 1;
    p.tiles_m = m / 256; p.tiles_n = n / 256; p.raster_group = 4; p.partial = (float *)st;
    auto kfn = gemm_b16_w4_kernel<false>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kfn, dim3(tiles), dim3(256), 128 * 1024, 0, p);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kfn, dim3(tiles), dim3(256), 128 * 1024, 0, p);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> s((size_t)tiles * 4 * 3);
    hipMemcpy(s.data(), st, s.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0, wt = 0;
    for (int i = 0; i < tiles * 4; ++i) { cyc += s[i * 3]; rt += s[i * 3 + 1]; wt += s[i * 3 + 2]; }
    cyc /= tiles * 4; rt /= tiles * 4; wt /= tiles * 4;
    printf("k %d: %.1f us per launch; loop %.0f shader cycles, %.0f realtime ticks (100 MHz) = %.1f us -> clock %.3f GHz; per k step %.0f cycles (MFMA pipe time 2048), of which %.0f at vmcnt(0) + barrier (%.1f %%)\n",
           k, ms * 50, cyc, rt, rt / 100.0, cyc / rt / 10.0, cyc / (k / 64), wt / (k / 64), 100.0 * wt / cyc);
    return 0;
}
