// Micro-benchmark for the bf16-exact policy's inner loop (development aid, not part of the product): e4m3 bytes are
// up-converted to bf16 in registers (v_cvt_scalef32_pk_bf16_fp8, exact), one 128-wide scale block = four chained
// v_mfma_f32_16x16x32_bf16 per 16x16 tile (or eight v_mfma_f32_32x32x16_bf16 per 32x32 tile), then the fp32 promotion.
// What it answers: cycles per MFMA with the conversions and the promotion beside it, at one and two waves per SIMD --
// i.e. whether the loop is bound by the matrix pipe or by vector issue, and which MFMA shape leaves more room.
// Register traffic is modelled on the 64(m) x 128(n) wave tile: a "pass" is half a k block (4 n-tiles of 16 / 2 of 32 against
// 4 m-tiles of 16 / 2 of 32); every conversion writes a register that a LATER MFMA reads (the other B set / the other A set),
// as in the kernel, so that no conversion is dead code and none feeds the MFMA right behind it.
//   CVT: conversions on (1.5 per 16x16x32-equivalent MFMA)   FMA: promotion on (1 v_fma_f32 per equivalent MFMA)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <bool HI>
__device__ __forceinline__ int cvt2(int raw)
{
    return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(raw, 1.0f, HI));
}

template <int SHAPE, int CVT, int FMA>
__global__ void __launch_bounds__(512) k(const int *seed, float *out, unsigned long long *cyc, int iters)
{
    int raw[8];
    for (int i = 0; i < 8; ++i) raw[i] = seed[(threadIdx.x * 8 + i) & 4095];
    const float s = 1.0001f;
    unsigned long long t0 = 0, t1 = 0;
    float r = 0;
    if constexpr (SHAPE == 0) {
        // 16 tiles per pass = 4 n-tiles x 4 m-tiles, 4 chained MFMAs each.  a[set][mt][q], b[set][q]: 8 bf16 per lane each.
        v4i a[2][4][4], b[2][4];
        for (int x = 0; x < 2; ++x)
            for (int q = 0; q < 4; ++q) {
                for (int j = 0; j < 4; ++j) b[x][q][j] = (j & 1) ? cvt2<true>(raw[(q + j + x) & 7]) : cvt2<false>(raw[(q + j + x) & 7]);
                for (int mt = 0; mt < 4; ++mt)
                    for (int j = 0; j < 4; ++j) a[x][mt][q][j] = (j & 1) ? cvt2<true>(raw[(mt + q + j) & 7]) : cvt2<false>(raw[(mt + q + j + x) & 7]);
            }
        v4f acc[16], part[2];
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
        part[0] = part[1] = v4f{0, 0, 0, 0};
        // one pass: MFMAs read A set `ar`, conversions fill A set `aw` (8 per n-tile: half an m-tile fragment) and, per
        // n-tile, the B set the NEXT n-tile reads (16)
        auto pass = [&](v4i (&ar)[4][4], v4i (&aw)[4][4]) {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int t = nt * 4 + mt;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (!CVT) asm volatile("" : "+v"(ar[mt][q]));   // opaque: the loop-invariant MFMA is not hoisted
                        // without the promotion the chain accumulates straight into the tile's accumulator (a plain bf16 GEMM)
                        const v4f c = q == 0 ? (FMA ? v4f{0, 0, 0, 0} : acc[t]) : part[t & 1];
                        part[t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, b[nt & 1][q]),
                                                                              __builtin_bit_cast(v8bf, ar[mt][q]), c, 0, 0, 0);
                        if (!FMA && q == 3) acc[t] = part[t & 1];
                        __builtin_amdgcn_sched_barrier(0);
                        if (CVT) {
                            // MFMA index inside the n-tile: u = 0..15; 24 conversions ride on them (1.5 per MFMA)
                            const int u = mt * 4 + q;
#pragma unroll
                            for (int cidx = (u * 24) / 16; cidx < ((u + 1) * 24) / 16; ++cidx) {
                                asm volatile("" : "+v"(raw[cidx & 7]));
                                if (cidx < 16) {   // the next n-tile's B fragment
                                    b[(nt & 1) ^ 1][cidx >> 2][cidx & 3] = (cidx & 1) ? cvt2<true>(raw[cidx & 7]) : cvt2<false>(raw[cidx & 7]);
                                } else {           // half of m-tile (nt >> 1 ...)'s next A fragment
                                    const int e = (nt & 1) * 8 + (cidx - 16);
                                    aw[nt][e >> 2][e & 3] = (cidx & 1) ? cvt2<true>(raw[cidx & 7]) : cvt2<false>(raw[cidx & 7]);
                                }
                            }
                        }
                        if (FMA) {   // promotion of the previous tile
                            const int j = (t + 15) & 15;
                            acc[j][q] = __builtin_fmaf(part[j & 1][q], s, acc[j][q]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; it += 2) {
            pass(a[0], a[1]);
            pass(a[1], a[0]);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
        r += part[0].x + part[1].x;
    } else {
        // 4 tiles of 32x32 per pass = 2 n-tiles x 2 m-tiles, 8 chained MFMAs each (the same flops as 16 tiles of 16x16).
        // One m-tile's fragment set is used for both m-tiles (the register budget of two 32-register A sets + two B sets).
        v4i a[2][8], b[2][8];
        for (int x = 0; x < 2; ++x)
            for (int q = 0; q < 8; ++q)
                for (int j = 0; j < 4; ++j) {
                    b[x][q][j] = (j & 1) ? cvt2<true>(raw[(q + j + x) & 7]) : cvt2<false>(raw[(q + j + x) & 7]);
                    a[x][q][j] = (j & 1) ? cvt2<true>(raw[(q + j + 3) & 7]) : cvt2<false>(raw[(q + j + x + 3) & 7]);
                }
        v16f acc[4], part[2];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int j = 0; j < 16; ++j) { part[0][j] = 0.f; part[1][j] = 0.f; }
        auto pass = [&](v4i (&ar)[8], v4i (&aw)[8]) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int t = nt * 2 + mt;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        asm volatile("" : "+v"(ar[q]));   // opaque: the two m-tiles (one fragment set) are not merged
                        v16f c;
                        if (q == 0) { if (FMA) { for (int j = 0; j < 16; ++j) c[j] = 0.f; } else c = acc[t]; } else c = part[t & 1];
                        part[t & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(v8bf, b[nt & 1][q]),
                                                                              __builtin_bit_cast(v8bf, ar[q]), c, 0, 0, 0);
                        if (!FMA && q == 7) acc[t] = part[t & 1];
                        __builtin_amdgcn_sched_barrier(0);
                        if (CVT) {
                            // MFMA index inside the n-tile: u = 0..15; 48 conversions ride on them (3 per MFMA = 1.5 per equivalent)
                            const int u = mt * 8 + q;
#pragma unroll
                            for (int cidx = u * 3; cidx < (u + 1) * 3; ++cidx) {
                                asm volatile("" : "+v"(raw[cidx & 7]));
                                if (cidx < 32) {
                                    b[(nt & 1) ^ 1][cidx >> 2][cidx & 3] = (cidx & 1) ? cvt2<true>(raw[cidx & 7]) : cvt2<false>(raw[cidx & 7]);
                                } else {
                                    const int e = nt * 16 + (cidx - 32);
                                    aw[e >> 2][e & 3] = (cidx & 1) ? cvt2<true>(raw[cidx & 7]) : cvt2<false>(raw[cidx & 7]);
                                }
                            }
                        }
                        if (FMA) {
                            const int j = (t + 3) & 3;
                            acc[j][2 * q] = __builtin_fmaf(part[j & 1][2 * q], s, acc[j][2 * q]);
                            acc[j][2 * q + 1] = __builtin_fmaf(part[j & 1][2 * q + 1], s, acc[j][2 * q + 1]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        };
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; it += 2) {
            pass(a[0], a[1]);
            pass(a[1], a[0]);
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
        r += part[0][0] + part[1][0];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int SHAPE, int CVT, int FMA>
static void run(const char *name, const int *seed, float *out, unsigned long long *cyc, int iters, int reps)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {256, 512}) {
        float ms = 0;
        for (int rep = 0; rep < reps; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL((k<SHAPE, CVT, FMA>), dim3(256), dim3(threads), 0, 0, seed, out, cyc, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        const int waves = 256 * threads / 64;
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        // per pass and wave: 64 MFMAs of 16x16x32 or 32 of 32x32x16 = 64 equivalents = 2 * 16*16*32 * 64 flops
        const double flops = 2.0 * 16 * 16 * 32 * 64.0 * iters * waves;
        const double eq = 64.0 * iters * (threads / 256);   // equivalents issued on one SIMD
        printf("%-36s waves/SIMD %d: %8.3f ms %6.0f TFLOP/s   %5.2f cyc per 16x16x32-equivalent per SIMD (16 = the pipe)   loop clock %.2f GHz\n",
               name, threads / 256, ms, flops / ms / 1e9, h[waves / 2] / eq, h[waves / 2] / (ms * 1e6));
    }
}

int main(int argc, char **argv)
{
    const int iters = 1000;
    const int reps = argc > 1 ? atoi(argv[1]) : 200;   // launches per case; the LAST is reported (sustained clocks)
    int *seed; float *out; unsigned long long *cyc;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<int> h(4096);
    srand(7);
    for (int i = 0; i < 4096; ++i) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) {
            unsigned byte = rand() & 0xFF;
            if ((byte & 0x7F) == 0x7F) byte &= 0x80;
            if ((byte & 0x78) > 0x58) byte &= 0xDF;
            v |= byte << (8 * b);
        }
        h[i] = (int)v;
    }
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 0, 0>("16x16x32 MFMA only", seed, out, cyc, iters, reps);
    run<0, 0, 1>("16x16x32 + promotion", seed, out, cyc, iters, reps);
    run<0, 1, 0>("16x16x32 + 1.5 cvt", seed, out, cyc, iters, reps);
    run<0, 1, 1>("16x16x32 + 1.5 cvt + promotion", seed, out, cyc, iters, reps);
    run<1, 0, 0>("32x32x16 MFMA only", seed, out, cyc, iters, reps);
    run<1, 0, 1>("32x32x16 + promotion", seed, out, cyc, iters, reps);
    run<1, 1, 0>("32x32x16 + 1.5 cvt", seed, out, cyc, iters, reps);
    run<1, 1, 1>("32x32x16 + 1.5 cvt + promotion", seed, out, cyc, iters, reps);
    return 0;
}
