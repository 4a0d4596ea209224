// Stamped build of the bf16-exact policy's masked-grouped kernel (gemm_fp8_bf16x_grouped_kernel.hpp; development aid): where the waves
// of a workgroup spend a k block, by wave half (waves 0-3 hold rows 0..63 of an expert, waves 4-7 rows 64..127).  Segments
// (s_memtime ticks per k block):  wait = the counted vmcnt wait for this block's pieces, barrier, head = first fragments out of the
// LDS + the refill issue + the first conversions, pipe = the MFMA / conversion / promotion loop, other = block bookkeeping + stores.
// usage: stamp_grouped_bx G N K rows [warm launches]      rows >= 0: masked_m[g] = rows;  rows < 0: random in [0, 128], seed -rows
#define DGA_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "gemm_fp8_bf16x_grouped_kernel.hpp"
using namespace dga;
#ifndef CFG_BNT
#define CFG_BNT true
#endif
int main(int argc, char **argv)
{
    const int groups = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 2048, k = argc > 3 ? atoi(argv[3]) : 7168;
    const int rows = argc > 4 ? atoi(argv[4]) : 128, warm = argc > 5 ? atoi(argv[5]) : 100, m = 128;
    typedef GemmCfg<128, 256, 2, 4, 3> Cfg;
    GemmParams p{};
    std::vector<uint8_t> ha((size_t)m * k), hb((size_t)n * k);
    srand(1);
    for (auto &v : ha) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0x80; if ((v & 0x78) > 0x60) v &= 0xBF; }
    for (auto &v : hb) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0x80; if ((v & 0x78) > 0x60) v &= 0xBF; }
    const int kb = (k + 127) / 128, nb = (n + 127) / 128;
    std::vector<float> hsa((size_t)m * kb, 1.0f), hsb((size_t)nb * kb, 0.5f);
    uint8_t *a, *b; float *sfa, *sfb; uint16_t *out; unsigned long long *st;
    hipMalloc(&a, (size_t)groups * m * k); hipMalloc(&b, (size_t)groups * n * k); hipMalloc(&sfa, hsa.size() * 4); hipMalloc(&sfb, hsb.size() * 4);
    hipMalloc(&out, (size_t)groups * m * n * 2);
    for (int g = 0; g < groups; ++g) {
        hipMemcpy(a + (size_t)g * m * k, ha.data(), ha.size(), hipMemcpyHostToDevice);
        hipMemcpy(b + (size_t)g * n * k, hb.data(), hb.size(), hipMemcpyHostToDevice);
    }
    hipMemcpy(sfa, hsa.data(), hsa.size() * 4, hipMemcpyHostToDevice); hipMemcpy(sfb, hsb.data(), hsb.size() * 4, hipMemcpyHostToDevice);
    p.a = a; p.sfa = sfa; p.b = b; p.sfb = sfb; p.out = out; p.m = m; p.n = n; p.k = k; p.kb_n = kb; p.nb_n = nb;
    p.lda = k; p.ldb = k; p.ldc = n; p.sfa_ld = kb; p.splitk = 1;
    p.tiles_m = 1; p.tiles_n = (n + Cfg::kBN - 1) / Cfg::kBN;
    p.groups = groups; p.b_groups = groups; p.a_gs = (int64_t)m * k; p.sfa_gs = 0; p.sfb_gs = 0; p.b_gs = (int64_t)n * k; p.c_gs = (int64_t)m * n;
    std::vector<int> hm(groups, rows);
    if (rows < 0) { std::mt19937 rng(-rows); for (auto &v : hm) v = rng() % 129; }
    long live = 0; for (int v : hm) live += v > 0;
    int *dm; hipMalloc(&dm, groups * 4); hipMemcpy(dm, hm.data(), groups * 4, hipMemcpyHostToDevice);
    p.masked_m = dm;
    p.raster_group = 1; p.xcd_remap = 1; p.b_nt = 1; p.out_nt = 1;
    const int tiles = groups * p.tiles_n, grid = std::min(tiles, 256), waves = 8;
    hipMalloc(&st, (size_t)grid * waves * 8 * 8); hipMemset(st, 0, (size_t)grid * waves * 8 * 8);
    p.stamps = st;
    auto kfn = gemm_fp8_bf16x_grouped_kernel<false, CFG_BNT>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * waves * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const double blocks = (double)live * p.tiles_n * kb / grid;   // k blocks a workgroup walks (average)
    printf("masked grouped %d x (rows %d, %d, %d) on the grouped kernel: %.1f us per launch (stamped build, %d warm launches), %.0f k blocks per workgroup\n",
           groups, rows, n, k, ms * 1000 / 20, warm, blocks);
    double ct = 0, crt = 0;
    for (int half = 0; half < 2; ++half) {
        double sg[8] = {0};
        for (int w = 0; w < grid * waves; ++w)
            if (((w % waves) >= 4) == half) {
                for (int q = 0; q < 6; ++q) sg[q] += (double)h[(size_t)w * 8 + q];
                ct += (double)h[(size_t)w * 8 + 6]; crt += (double)h[(size_t)w * 8 + 7];
            }
        const double hw = (double)grid * 4 * blocks;
        printf("  waves %s per k block: wait %.0f  barrier %.0f  head %.0f  pipe %.0f  other %.0f  = %.0f ticks\n", half ? "4-7" : "0-3",
               sg[1] / hw, sg[2] / hw, sg[3] / hw, sg[4] / hw, (sg[0] + sg[5]) / hw, (sg[0] + sg[1] + sg[2] + sg[3] + sg[4] + sg[5]) / hw);
    }
    printf("  kernel: %.0f ticks per wave, clock %.3f GHz, %.1f us in the loop\n", ct / (grid * waves), ct / crt * 0.1, crt / (grid * waves) / 100.0);
    return 0;
}
