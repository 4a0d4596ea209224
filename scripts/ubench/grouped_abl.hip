// Ablation harness for the masked grouped stream (BASELINE configs[3]: 256 experts x (128, 7168, 2048), full mask):
// what keeps the weight stream below the copy rate?  One binary per -D switch set (Makefile: grouped_abl_*), each
// prints the launch time and the GB/s of algorithmic bytes.  Switches (gemm_fp8_kernel.hpp): DGA_ABL_NOMFMA (no matrix
// work), DGA_ABL_NOADMA (the A tile is not fetched), DGA_ABL_NOSTORE, DGA_ABL_NOFMA, DGA_ABL_NOLDS; CFG_* picks the build.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_fp8_kernel.hpp"
using namespace dga;
#ifndef CFG_BM
#define CFG_BM 128
#define CFG_BN 256
#define CFG_WM 2
#define CFG_WN 2
#define CFG_ST 3
#endif
#ifndef CFG_LC
#define CFG_LC 0
#endif
int main(int argc, char **argv)
{
    const int G = argc > 1 ? atoi(argv[1]) : 256, reps = argc > 2 ? atoi(argv[2]) : 20;
    // optional: m n k of ONE group's problem (default the masked grouped configs[3]; "1 40 name 4096 2048 7168" = dense configs[2])
    const int m = argc > 4 ? atoi(argv[4]) : 128, n = argc > 5 ? atoi(argv[5]) : 2048, k = argc > 6 ? atoi(argv[6]) : 7168;
    typedef GemmCfg<CFG_BM, CFG_BN, CFG_WM, CFG_WN, CFG_ST, CFG_LC> Cfg;
    const int kb = k / 128, nb = n / 128;
    const size_t abytes = (size_t)G * m * k, bbytes = (size_t)G * n * k;
    std::vector<uint8_t> h(1 << 24);
    srand(1);
    for (auto &v : h) { v = rand() & 0xFF; if ((v & 0x7F) == 0x7F) v &= 0x80; }
    uint8_t *a, *b; float *sfa, *sfb; uint16_t *out; int32_t *mm;
    hipMalloc(&a, abytes); hipMalloc(&b, bbytes); hipMalloc(&sfa, (size_t)G * m * kb * 4); hipMalloc(&sfb, (size_t)G * nb * kb * 4);
    hipMalloc(&out, (size_t)G * m * n * 2); hipMalloc(&mm, G * 4);
    for (size_t o = 0; o < abytes; o += h.size()) hipMemcpy(a + o, h.data(), std::min(h.size(), abytes - o), hipMemcpyHostToDevice);
    for (size_t o = 0; o < bbytes; o += h.size()) hipMemcpy(b + o, h.data(), std::min(h.size(), bbytes - o), hipMemcpyHostToDevice);
    std::vector<float> ones((size_t)G * m * kb, 1.0f);
    hipMemcpy(sfa, ones.data(), ones.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(sfb, ones.data(), (size_t)G * nb * kb * 4, hipMemcpyHostToDevice);
    std::vector<int32_t> hm(G, m);
    hipMemcpy(mm, hm.data(), G * 4, hipMemcpyHostToDevice);
    GemmParams p{};
    p.a = a; p.sfa = sfa; p.b = b; p.sfb = sfb; p.out = out; p.masked_m = mm; p.m = m; p.n = n; p.k = k; p.kb_n = kb; p.nb_n = nb;
    p.lda = k; p.ldb = k; p.ldc = n; p.groups = G; p.b_groups = G; p.sfa_ld = kb;
    p.a_gs = (int64_t)m * k; p.b_gs = (int64_t)n * k; p.c_gs = (int64_t)m * n; p.sfa_gs = (int64_t)m * kb; p.sfb_gs = (int64_t)nb * kb;
#ifdef INDEXED   // the indexed form on an identity row table: the same bytes at the same addresses as the packed layout
    {
        std::vector<int64_t> hi((size_t)G * m);
        for (size_t i = 0; i < hi.size(); ++i) hi[i] = (int64_t)i;
        int64_t *ri; hipMalloc(&ri, hi.size() * 8);
        hipMemcpy(ri, hi.data(), hi.size() * 8, hipMemcpyHostToDevice);
        p.row_index = ri; p.a_gs = p.c_gs = p.sfa_gs = 0; p.a_bytes = (int64_t)G * m * k;
    }
#endif
    p.tiles_m = (m + Cfg::kBM - 1) / Cfg::kBM; p.tiles_n = (n + Cfg::kBN - 1) / Cfg::kBN;
    p.raster_group = p.tiles_m >= 4 ? 4 : 1; p.xcd_remap = 1; p.splitk = 1;
    // optional 7th argument: operand sets rotated launch by launch (cold-cache protocol for shapes that fit the Infinity Cache)
    const int sets = argc > 7 ? atoi(argv[7]) : 1;
    std::vector<GemmParams> ps(1, p);
    for (int sidx = 1; sidx < sets; ++sidx) {
        GemmParams q = p;
        uint8_t *b2; hipMalloc(&b2, bbytes);
        hipMemcpy(b2, b, bbytes, hipMemcpyDeviceToDevice);
        q.b = b2;
        ps.push_back(q);
    }
    const int grid = G * p.tiles_m * p.tiles_n;
    auto kfn = gemm_fp8_blockscaled_nt_kernel<Cfg, 0, false>;
    hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, ps[i % sets]);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(Cfg::NT), Cfg::LDS_BYTES, 0, ps[i % sets]);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1000 / reps;
    const double bytes = (double)bbytes + (double)abytes + (double)G * m * (4.0 * kb + 2.0 * n) + (double)G * nb * kb * 4;
    printf("%-28s tile %dx%d waves %dx%d%s stages %d: %8.1f us  %7.1f GB/s algorithmic (weights alone %7.1f GB/s)  err=%d\n",
           argc > 3 ? argv[3] : "", Cfg::kBM, Cfg::kBN, Cfg::kWM, Cfg::kWN, Cfg::kLC ? "+loaders" : "", Cfg::STAGES, us, bytes / us / 1e3, bbytes / us / 1e3,
           (int)hipGetLastError());
    return 0;
}
