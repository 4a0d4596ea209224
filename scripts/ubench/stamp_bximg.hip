// Stamped build of the bf16-exact policy's image kernel (csrc/gemm_fp8_bf16x_image_kernel.hpp) on a dense problem (development
// aid): where a wave spends a k block.  The DGA_BXI_* switches of the kernel remove one cost each.  Segments (s_memtime ticks per
// k block, averaged over all waves):
//   head   gaps 0..35   (tile row 0: the block's B fragments stream in; the A pieces of the next image are converted / stored)
//   Y      lgkmcnt(0) + barrier "every wave holds the block's B fragments"
//   body   gaps 36..111 (the B pieces of the next image are converted / stored)
//   X      lgkmcnt(0) + barrier "the next block's images are complete"
//   tail   gaps 112..127 (the next block's first fragments are read)
// usage: stamp_bximg M N K [warm launches]
#define DGA_STAMPS 1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_fp8_bf16x_aimage_kernel.hpp"
using namespace dga;
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 4096, n = argc > 2 ? atoi(argv[2]) : 4096, k = argc > 3 ? atoi(argv[3]) : 4096;
    const int warm = argc > 4 ? atoi(argv[4]) : 2000;
#ifndef BXI_WAVES
#define BXI_WAVES 8
#endif
#ifdef BXI_AIMAGE   // the A-image build (gemm_fp8_bf16x_aimage_kernel.hpp): head = gaps 0..31, X, tail = gaps 32..63 (Y, body: none)
    typedef BxAImageCfg Cfg;
#else
    typedef BxImageCfg<BXI_WAVES> Cfg;
#endif
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
    p.lda = k; p.ldb = k; p.ldc = n; p.groups = 1; p.b_groups = 1; p.sfa_ld = kb; p.splitk = 1;
    p.tiles_m = (m + Cfg::kBM - 1) / Cfg::kBM; p.tiles_n = (n + Cfg::kBN - 1) / Cfg::kBN;
    p.raster_group = 4; p.xcd_remap = 1;
    const int grid = p.tiles_m * p.tiles_n, waves = Cfg::NT / 64;
    hipMalloc(&st, (size_t)grid * waves * 8 * 8); hipMemset(st, 0, (size_t)grid * waves * 8 * 8);
    p.stamps = st;
#ifdef BXI_AIMAGE
    auto kfn = gemm_fp8_bf16x_aimage_kernel<false>;
#else
    auto kfn = gemm_fp8_bf16x_image_kernel<Cfg, false>;
#endif
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
        for (int q = 0; q < 6; ++q) seg[q] += (double)h[(size_t)w * 8 + q];
        ct += (double)h[(size_t)w * 8 + 6]; crt += (double)h[(size_t)w * 8 + 7];
    }
    const double nw = (double)grid * waves;
    printf("image build 128x256, %d waves, on %dx%dx%d: %.1f us per launch (stamped build, %d warm launches)\n", Cfg::NT / 64, m, n, k, ms * 1000 / 20, warm);
    printf("  per k block: head %.0f  Y %.0f  body %.0f  X %.0f  tail %.0f  = %.0f ticks (matrix pipe alone: 2048 per SIMD)\n", seg[0] / nw / kb,
           seg[1] / nw / kb, seg[2] / nw / kb, seg[3] / nw / kb, seg[4] / nw / kb, (seg[0] + seg[1] + seg[2] + seg[3] + seg[4]) / nw / kb);
    printf("  whole wave: %.0f ticks, clock %.3f GHz, %.1f us\n", ct / nw, ct / crt * 0.1, crt / nw / 100.0);
    return 0;
}
