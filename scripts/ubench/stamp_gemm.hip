// Diagnostic build of the fp8 GEMM kernel with s_memtime stamps (shares, not run time, are meaningful).
#define DGA_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../deepgemm_ascend_amd/csrc/gemm_fp8_kernel.hpp"
using namespace dga;
__device__ uint8_t zero_chunk[256];
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 4096, n = argc > 2 ? atoi(argv[2]) : 4096, k = argc > 3 ? atoi(argv[3]) : 4096;
    typedef GemmCfg<256, 256, 4, 2> Cfg;
    GemmParams p{};
    std::vector<uint8_t> ha((size_t)m * k), hb((size_t)n * k);
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
    p.lda = k; p.ldb = k; p.ldc = n; p.groups = 1; p.tiles_m = (m + 255) / 256; p.tiles_n = (n + 255) / 256;
    p.raster_group = 8; p.xcd_remap = 1;
    const int grid = p.tiles_m * p.tiles_n;
    hipMalloc(&st, (size_t)grid * 8 * 8 * 8); hipMemset(st, 0, (size_t)grid * 8 * 8 * 8);
    p.stamps = st;
    const int pp = argc > 4 ? atoi(argv[4]) : 0;
#ifndef CFG_MATH
#define CFG_MATH 0     // 2: the hardware-scale build (block scales in the MFMA's E8M0 operands; the scales below are powers of two)
#endif
    auto kfn = pp == 2 ? gemm_fp8_blockscaled_nt_kernel<Cfg, 2, false, false, CFG_MATH>
             : pp ? gemm_fp8_blockscaled_nt_kernel<Cfg, 1, false> : gemm_fp8_blockscaled_nt_kernel<Cfg, 0, false, false, CFG_MATH>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = argc > 5 ? atoi(argv[5]) : 5;   // 4000+ = sustained clocks (the default 5 is inside the ramp after idle)
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("stamped kernel (pp=%d): %.1f us per launch (%d x %d x %d)\n", pp, ms * 1000 / 20, m, n, k);
    std::vector<unsigned long long> h((size_t)grid * 8 * 8);
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    const char *names0[6] = {"issue next-stage DMA", "vmcnt wait", "barrier stage-ready", "first frags from LDS", "MFMA+promotion pipeline", "barrier stage-free"};
    const char *names1[6] = {"wait + barrier Ba", "read A frags + scales", "H1 (+DMA)", "wait + barrier Bb", "H2 (+DMA)", "-"};
    const char **names = pp ? names1 : names0;
    for (int half = 0; half < 2; ++half) {
        double tot = 0, seg[6] = {0};
        for (int w = 0; w < grid * 8; ++w) {
            if (((w % 8) >= 4) != half) continue;
            for (int q = 0; q < 6; ++q) { seg[q] += h[(size_t)w * 8 + q]; tot += h[(size_t)w * 8 + q]; }
        }
        printf("waves %s: per k-step cycles (s_memtime ticks), share\n", half ? "4-7" : "0-3");
        for (int q = 0; q < 6; ++q) printf("  %-26s %8.0f  %5.1f%%\n", names[q], seg[q] / (grid * 4) / kb, 100 * seg[q] / tot);
        printf("  total per k-step %.0f\n", tot / (grid * 4) / kb);
    }
    double ct = 0, crt = 0;
    for (int w = 0; w < grid * 8; ++w) { ct += (double)h[(size_t)w * 8 + 6]; crt += (double)h[(size_t)w * 8 + 7]; }
    printf("main loop: %.0f shader ticks, %.0f realtime ticks (100 MHz) per wave -> clock %.3f GHz, loop %.1f us\n",
           ct / (grid * 8), crt / (grid * 8), ct / crt * 0.1, crt / (grid * 8) / 100.0);
    if (pp == 2) {   // absolute real-time stamps of the last launch: entry / loop start / loop end / stores issued
        double pro = 0, loop = 0, epi = 0;
        unsigned long long first = ~0ull, last = 0, first_loop_end = ~0ull, last_loop_start = 0;
        for (int w = 0; w < grid * 8; ++w) {
            const unsigned long long *q = &h[(size_t)w * 8];
            pro += (double)(q[1] - q[0]); loop += (double)(q[2] - q[1]); epi += (double)(q[3] - q[2]);
            first = std::min(first, q[0]); last = std::max(last, q[3]);
            first_loop_end = std::min(first_loop_end, q[2]); last_loop_start = std::max(last_loop_start, q[1]);
        }
        const double nw = grid * 8.0;
        printf("per wave: entry -> loop %.2f us, loop %.2f us, loop end -> stores issued %.2f us; first entry -> last exit %.2f us "
               "(last loop start at +%.2f, first loop end at +%.2f); launch interval %.2f us\n",
               pro / nw / 100, loop / nw / 100, epi / nw / 100, (double)(last - first) / 100, (double)(last_loop_start - first) / 100,
               (double)(first_loop_end - first) / 100, ms * 1000 / 20);
    }
    return 0;
}
