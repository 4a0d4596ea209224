// Probe: do buffer_load_dwordx4 (to registers) and buffer_load_dwordx4 ... lds (LDS-DMA) accept source addresses at ANY byte
// alignment on gfx950, and what do they cost?  (The odd-K operands of the fp8 GEMM start their rows at arbitrary bytes.)
// For each misalignment 0..15: correctness of both paths against the host copy, then a bandwidth run of each path at
// misalignment 0 / 1 / 4 / 8 over a 256 MB buffer.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "dga_device_common.hpp"
using namespace dga;

__global__ void __launch_bounds__(256) copy_regs(const uint8_t *src, uint8_t *dst, uint32_t mis, uint32_t bytes)
{
    const v4i rs = make_rsrc(src, bytes + 64);
    const uint32_t off = (blockIdx.x * 256 + threadIdx.x) * 16;
    if (off >= bytes) return;
    v4i r;
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(off + mis), "s"(rs) : "memory");
    *(v4i *)(dst + off) = r;
}
__global__ void __launch_bounds__(256) copy_dma(const uint8_t *src, uint8_t *dst, uint32_t mis, uint32_t bytes)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4096];
    const v4i rs = make_rsrc(src, bytes + 64);
    const uint32_t off = (blockIdx.x * 256 + threadIdx.x) * 16;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)lds;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    dma16(off < bytes ? off + mis : 0x80000000u, rs, 0, lds0 + wave * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (off < bytes) *(v4i *)(dst + off) = *(const v4i *)(lds + threadIdx.x * 16);
}
int main()
{
    const uint32_t bytes = 256u << 20;
    uint8_t *src, *dst;
    hipMalloc(&src, bytes + 4096); hipMalloc(&dst, bytes);
    std::vector<uint8_t> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint8_t)(i * 131 + (i >> 8) * 7);
    for (uint32_t o = 0; o < bytes + 4096; o += (1 << 20)) hipMemcpy(src + o, h.data(), std::min<size_t>(1 << 20, bytes + 4096 - o), hipMemcpyHostToDevice);
    std::vector<uint8_t> got(1 << 16);
    for (int path = 0; path < 2; ++path)
        for (uint32_t mis = 0; mis < 16; ++mis) {
            hipMemset(dst, 0xEE, 1 << 16);
            if (path == 0) hipLaunchKernelGGL(copy_regs, dim3(16), dim3(256), 0, 0, src, dst, mis, 1u << 16);
            else hipLaunchKernelGGL(copy_dma, dim3(16), dim3(256), 0, 0, src, dst, mis, 1u << 16);
            hipDeviceSynchronize();
            hipMemcpy(got.data(), dst, 1 << 16, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (size_t i = 0; i < got.size(); ++i) bad += got[i] != h[(i + mis) % h.size()];
            printf("%s misalignment %2u: %zu of %zu bytes wrong\n", path ? "LDS-DMA  " : "registers", mis, bad, got.size());
        }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int path = 0; path < 2; ++path)
        for (uint32_t mis : {0u, 1u, 4u, 8u, 3u}) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (path == 0) hipLaunchKernelGGL(copy_regs, dim3(bytes / 4096), dim3(256), 0, 0, src, dst, mis, bytes);
                else hipLaunchKernelGGL(copy_dma, dim3(bytes / 4096), dim3(256), 0, 0, src, dst, mis, bytes);
                hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
            }
            printf("%s misalignment %2u: %.1f us for 256 MB read + 256 MB written = %.2f TB/s\n", path ? "LDS-DMA  " : "registers", mis, ms * 1e3,
                   2.0 * bytes / (ms * 1e-3) / 1e12);
        }
    return 0;
}
