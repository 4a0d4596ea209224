// Stamped build of ONE plain-loop tile build of the fp8 kernel on a dense problem (development aid): where the computing waves
// of e.g. the loader-wave 128x256 build (BASELINE configs[2]) spend a k block.  -DCFG_BM/BN/WM/WN/ST/LC pick the build, the
// DGA_ABL_* switches of gemm_fp8_kernel.hpp remove one cost.  Segments (s_memtime ticks per k block, computing waves only):
//   wait    vmcnt wait of the waves that issue their own DMA (loader-wave builds: 0)
//   barrier the k block's one barrier: "this stage landed everywhere" (in a loader-wave build this is where a late refill shows)
//   frags   first fragments + scales out of the LDS
//   pipe    the MFMA / promotion pipeline (+ B-fragment reads, + the refill issue in builds without loader waves)
// usage: stamp_tile M N K [warm launches] [split-K factor: the tile kernel of a two-launch split-K call, slabs written, no combine]
#define DGA_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_fp8_kernel.hpp"
using namespace dga;
#ifndef CFG_MATH
#define CFG_MATH 0      // 1: the bf16-exact loop (gemm_fp8_kernel.hpp MATH = 1; three stages, no loader waves)
#endif
#ifndef CFG_BM
#define CFG_BM 128
#define CFG_BN 256
#define CFG_WM 2
#define CFG_WN 2
#define CFG_ST 3
#define CFG_LC 4
#endif
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 4096, n = argc > 2 ? atoi(argv[2]) : 2048, k = argc > 3 ? atoi(argv[3]) : 7168;
    const int warm = argc > 4 ? atoi(argv[4]) : 3000;
    const int splitk = argc > 5 ? atoi(argv[5]) : 1;
    const int groups = argc > 6 ? atoi(argv[6]) : 1;      // masked grouped layout: `groups` problems of M x N x K ...
    const int rows = argc > 7 ? atoi(argv[7]) : m;        // ... with masked_m[g] = rows
    typedef GemmCfg<CFG_BM, CFG_BN, CFG_WM, CFG_WN, CFG_ST, CFG_LC> Cfg;
    GemmParams p{};
    std::vector<uint8_t> ha((size_t)m * k), hb((size_t)n * k * groups);
    srand(1);
    for (auto &v : ha) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0x80; if ((v & 0x78) > 0x60) v &= 0xBF; }
    for (auto &v : hb) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0x80; if ((v & 0x78) > 0x60) v &= 0xBF; }
    const int kb = (k + 127) / 128, nb = (n + 127) / 128;
    std::vector<float> hsa((size_t)m * kb, 1.0f), hsb((size_t)nb * kb, 0.5f);
    uint8_t *a, *b; float *sfa, *sfb; uint16_t *out; unsigned long long *st;
    hipMalloc(&a, ha.size()); hipMalloc(&b, hb.size()); hipMalloc(&sfa, hsa.size() * 4); hipMalloc(&sfb, hsb.size() * 4);
    hipMalloc(&out, (size_t)m * n * 2);
    hipMemcpy(a, ha.data(), ha.size(), hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), hb.size(), hipMemcpyHostToDevice);
    hipMemcpy(sfa, hsa.data(), hsa.size() * 4, hipMemcpyHostToDevice); hipMemcpy(sfb, hsb.data(), hsb.size() * 4, hipMemcpyHostToDevice);
    p.a = a; p.sfa = sfa; p.b = b; p.sfb = sfb; p.out = out; p.m = m; p.n = n; p.k = k; p.kb_n = kb; p.nb_n = nb;
    p.lda = k; p.ldb = k; p.ldc = n; p.groups = 1; p.b_groups = 1; p.sfa_ld = kb; p.splitk = 1;
    p.tiles_m = (m + Cfg::kBM - 1) / Cfg::kBM; p.tiles_n = (n + Cfg::kBN - 1) / Cfg::kBN;
    if (groups > 1) {   // every group has its own A, B and output (argv[8] = 1: one A for all -- L2-resident)
        const bool shared_a = argc > 8 && atoi(argv[8]) == 1;
        uint8_t *ag = a; uint16_t *og = out;
        if (!shared_a) {
            hipMalloc(&ag, (size_t)groups * m * k);
            for (int g = 0; g < groups; ++g) hipMemcpy(ag + (size_t)g * m * k, a, (size_t)m * k, hipMemcpyDeviceToDevice);
            hipMalloc(&og, (size_t)groups * m * n * 2);
            p.a = ag; p.out = og;
        }
        p.groups = groups; p.b_groups = groups; p.a_gs = shared_a ? 0 : (int64_t)m * k; p.sfa_gs = 0; p.sfb_gs = 0; p.b_gs = (int64_t)n * k;
        p.c_gs = shared_a ? 0 : (int64_t)m * n;
        std::vector<int> hm(groups, rows);
        int *dm; hipMalloc(&dm, groups * 4); hipMemcpy(dm, hm.data(), groups * 4, hipMemcpyHostToDevice);
        p.masked_m = dm;
    }
    p.raster_group = 4; p.xcd_remap = 1;
    const int kb_total = kb;
    if (splitk > 1) {
        p.splitk = splitk;
        p.kb_per_split = (kb + splitk - 1) / splitk;
        float *slab;
        hipMalloc(&slab, (size_t)splitk * m * n * 4);
        p.partial = slab;
    }
    const int kb_wave = splitk > 1 ? p.kb_per_split : kb_total;   // k blocks one wave walks
    const int grid = p.tiles_m * p.tiles_n * splitk * groups, waves = Cfg::NT / 64, cwaves = Cfg::kWM * Cfg::kWN;
    hipMalloc(&st, (size_t)grid * waves * 8 * 8); hipMemset(st, 0, (size_t)grid * waves * 8 * 8);
    p.stamps = st;
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, 0, false, false, CFG_MATH>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h((size_t)grid * waves * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    double seg[8] = {0}, ct = 0, crt = 0;
    for (int w = 0; w < grid * waves; ++w) {
        if (w % waves >= cwaves) continue;   // loader waves write no stamps
        for (int q = 0; q < 6; ++q) seg[q] += (double)h[(size_t)w * 8 + q];
        ct += (double)h[(size_t)w * 8 + 6]; crt += (double)h[(size_t)w * 8 + 7];
    }
    const double nw = (double)grid * cwaves;
    printf("tile %dx%d waves %dx%d%s stages %d on %dx%dx%d: %.1f us per launch (stamped build, %d warm launches)\n", Cfg::kBM, Cfg::kBN,
           Cfg::kWM, Cfg::kWN, Cfg::kLC ? "+loaders" : "", Cfg::STAGES, m, n, k, ms * 1000 / 20, warm);
    printf("  per k block: wait %.0f  barrier %.0f  frags %.0f  pipe %.0f  = %.0f ticks (matrix pipe alone: %d)\n", seg[1] / nw / kb_wave,
           seg[2] / nw / kb_wave, seg[3] / nw / kb_wave, seg[4] / nw / kb_wave, (seg[1] + seg[2] + seg[3] + seg[4]) / nw / kb_wave,
           Cfg::TM * Cfg::TN * (CFG_MATH ? 64 : 32) * cwaves / 4);
    printf("  main loop: %.0f ticks, clock %.3f GHz, %.1f us\n", ct / nw, ct / crt * 0.1, crt / nw / 100.0);
    if (cwaves == 8)   // the two waves of a SIMD: first-dispatched half against second-dispatched half
        for (int half = 0; half < 2; ++half) {
            double sg[8] = {0};
            for (int w = 0; w < grid * waves; ++w)
                if (w % waves < cwaves && ((w % waves) >= 4) == half)
                    for (int q = 0; q < 6; ++q) sg[q] += (double)h[(size_t)w * 8 + q];
            const double hw = (double)grid * 4;
            printf("  waves %s: wait %.0f  barrier %.0f  frags %.0f  pipe %.0f\n", half ? "4-7" : "0-3", sg[1] / hw / kb_wave, sg[2] / hw / kb_wave,
                   sg[3] / hw / kb_wave, sg[4] / hw / kb_wave);
        }
    return 0;
}
