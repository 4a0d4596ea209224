// How fast does a 256x256 bf16 tile per workgroup (256 workgroups, 33.5 MB in all, row stride 8192 B) reach memory, by the
// shape of one wave-level store instruction?  (development aid)
//   0  16 rows x 64 B   (the fp8 tile kernel's epilogue: lane = (row li, 16-byte column group kg))
//   1   8 rows x 128 B  (full cache lines per row)
//   2   4 rows x 256 B
//   3   2 rows x 512 B
// Each workgroup writes its tile with 8 waves x 16 instructions of 1 KiB.  Usage: store_patterns [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));

template <int PATTERN>
__global__ void __launch_bounds__(512) store_kernel(uint16_t *out, int ld, int tiles_n, int spin)
{
    const int tile = blockIdx.x, tm = tile / tiles_n, tn = tile % tiles_n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint16_t *base = out + (size_t)(tm * 256) * ld + tn * 256;
    // some work first so that all workgroups store at about the same time, as at the end of a GEMM launch
    float x = (float)threadIdx.x;
    for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;
    const v4i v = v4i{(int)x, lane, wave, tile};
    constexpr int ROWS = PATTERN == 0 ? 16 : PATTERN == 1 ? 8 : PATTERN == 2 ? 4 : 2;   // rows per wave-instruction
    static_assert(64 / ROWS <= 32, "a wave's block is 512 B wide");
    constexpr int LPR = 64 / ROWS;                                                   // lanes per row (16 B each)
    // the wave owns 32 rows x 256 columns (512 B per row) = 16 KiB = 16 instructions
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int chunk = it * 64 + lane;                 // 16-byte chunk index inside the wave's 32 x 512 B block
        int row, colb;
        if (PATTERN == 0) { row = (it >> 3) * 16 + (lane & 15); colb = (it & 7) * 64 + (lane >> 4) * 16; }
        else { const int per_row = 512 / 16; const int inst_rows = ROWS; const int r_in = lane / LPR, c_in = lane % LPR;
               const int blocks_per_row = per_row / LPR;   // instructions needed to cover a row span
               row = (it / blocks_per_row) * inst_rows + r_in; colb = ((it % blocks_per_row) * LPR + c_in) * 16; (void)chunk; }
        *(v4i *)((uint8_t *)(base + (size_t)(wave * 32 + row) * ld) + colb) = v;
    }
}

template <int P>
static void run(uint16_t *out, int reps, const char *name)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int spin : {0, 20000}) {
        for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(256), dim3(512), 0, 0, out, 4096, 16, spin);
        hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(store_kernel<P>, dim3(256), dim3(512), 0, 0, out, 4096, 16, spin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("pattern %d (%s) spin %5d: %7.2f us per launch%s\n", P, name, spin, ms * 1000 / reps, spin ? "" : "  (stores only)");
    }
}

int main(int argc, char **argv)
{
    const int reps = argc > 1 ? atoi(argv[1]) : 200;
    uint16_t *out; hipMalloc(&out, (size_t)4096 * 4096 * 2);
    for (int r = 0; r < 2; ++r) {
        run<0>(out, reps, "16 rows x 64 B");
        run<1>(out, reps, "8 rows x 128 B");
        run<2>(out, reps, "4 rows x 256 B");
        run<3>(out, reps, "2 rows x 512 B");
    }
    return 0;
}
