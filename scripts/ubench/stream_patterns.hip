// What read rate does the memory system give an LDS-DMA weight stream, by access pattern?  (development aid)
// 256 workgroups (one per CU) x 4 waves stream a [ROWS_TOTAL, K] byte matrix into a 3-stage LDS ring, 32 KB per step, nothing
// is computed.  Patterns:
//   0  tile rows: 128 B of each of 256 rows per step (row stride K)      -- the [N, K] weight layout, BK = 128
//   1  packed:    32 KB contiguous per step                               -- weights pre-packed in tile order
//   2  tile rows: 256 B of each of 128 rows per step                      -- BK = 256, BN = 128
//   3  tile rows: 512 B of each of 64 rows per step
//   4  pattern 0 with the k blocks of a tile visited in a per-workgroup rotated order (de-phased DRAM pages)
// Usage: stream_patterns <pattern> [stages_in_flight=2] [reps=20]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef __attribute__((address_space(3))) void *lptr_t;

__device__ __forceinline__ void dma16g(const void *src, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_addr) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int K = 7168, KB = K / 128, TILE_ROWS = 256, STAGE = 32768, ITERS = 8;

template <int PATTERN, int DEPTH>
__global__ void __launch_bounds__(256) stream_kernel(const uint8_t *w, int tiles, unsigned *sink)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t lds0 = (uint32_t)(uintptr_t)(lptr_t)smem;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = tiles / 8, step = gridDim.x / 8;
    int stage = 0, outstanding = 0;
    for (int local = slot; local < per_xcd; local += step) {
        const int tile = xcd * per_xcd + local;
        const uint8_t *base = w + (size_t)tile * TILE_ROWS * K;
        for (int kb0 = 0; kb0 < KB; ++kb0) {
            int kb = kb0;
            if (PATTERN == 4) { kb = kb0 + (blockIdx.x * 7) % KB; if (kb >= KB) kb -= KB; }
            // wait until at most DEPTH-1 steps are in flight, then everyone may overwrite the oldest stage
            if (outstanding == DEPTH) { wait_vmcnt<(DEPTH - 1) * ITERS>(); --outstanding; }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = it * 256 + tid;   // 16-byte chunk of the 32 KB step
                const uint8_t *src;
                if (PATTERN == 0 || PATTERN == 4) src = base + (size_t)(c >> 3) * K + kb * 128 + (c & 7) * 16;
                else if (PATTERN == 1) src = base + (size_t)kb * STAGE + (size_t)c * 16;
                else if (PATTERN == 2) src = base + (size_t)((kb & 1) * 128 + (c >> 4)) * K + (kb >> 1) * 256 + (c & 15) * 16;
                else src = base + (size_t)((kb & 3) * 64 + (c >> 5)) * K + (kb >> 2) * 512 + (c & 31) * 16;
                dma16g(src, lds0 + stage * STAGE + it * 4096 + wave * 1024);
            }
            ++outstanding;
            stage = stage + 1 == 3 ? 0 : stage + 1;
        }
    }
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (sink && tid == 0 && smem[blockIdx.x & 1023] == 0x5A && smem[7] == 0xA5) atomicAdd(sink, 1u);
}

template <int P, int D>
static void run(const uint8_t *w, int tiles, unsigned *sink, int reps, const char *name)
{
    auto kfn = stream_kernel<P, D>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * STAGE);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 3 * STAGE, 0, w, tiles, sink);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kfn, dim3(256), dim3(256), 3 * STAGE, 0, w, tiles, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / reps, bytes = (double)tiles * TILE_ROWS * K;
    printf("pattern %d (%s) depth %d: %8.1f us  %7.1f GB/s  err=%d\n", P, name, D, us, bytes / us / 1e3, (int)hipGetLastError());
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    const int tiles = 2048;                       // 256 experts x 8 tiles of 256 rows: 3.758 GB
    const size_t bytes = (size_t)tiles * TILE_ROWS * K;
    uint8_t *w; unsigned *sink;
    hipMalloc(&w, bytes); hipMalloc(&sink, 4);
    hipMemset(w, 0x11, bytes); hipMemset(sink, 0, 4);
    for (int round = 0; round < 2; ++round) {
        run<0, 2>(w, tiles, sink, reps, "128 B x 256 rows");
        run<1, 2>(w, tiles, sink, reps, "32 KB contiguous");
        run<2, 2>(w, tiles, sink, reps, "256 B x 128 rows");
        run<3, 2>(w, tiles, sink, reps, "512 B x 64 rows");
        run<4, 2>(w, tiles, sink, reps, "128 B x 256 rows, rotated k");
        run<0, 3>(w, tiles, sink, reps, "128 B x 256 rows");
        run<1, 3>(w, tiles, sink, reps, "32 KB contiguous");
    }
    return 0;
}
