// Issue cost of the candidate e4m3 -> 16-bit up-conversion instructions on gfx950 (development aid, not part of the product):
// cycles per wave-instruction of an unrolled stream of independent instructions, one and two waves per SIMD, alone and riding
// behind one v_mfma_f32_16x16x32_bf16 each N instructions.  Answers: what one conversion costs the vector issue port, and how
// many fit behind an MFMA (16 cycles of matrix pipe) before the loop turns issue-bound.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));

template <int OP>
__device__ __forceinline__ void one(int &d, int &d2, int s, float sc)
{
    if constexpr (OP == 0) asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, %2" : "=v"(d) : "v"(s), "v"(sc));
    if constexpr (OP == 1) asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %1, %2" : "=v"(d) : "v"(s), "v"(sc));
    if constexpr (OP == 2) { long long r; asm volatile("v_cvt_pk_f32_fp8 %0, %1" : "=v"(r) : "v"(s)); d = (int)r; d2 = (int)(r >> 32); }
    if constexpr (OP == 3) asm volatile("v_perm_b32 %0, %1, %1, %2" : "=v"(d) : "v"(s), "v"(0x0c010c00));
    if constexpr (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %1" : "=v"(d) : "v"(s), "v"(sc));
    if constexpr (OP == 5) asm volatile("v_pk_ashrrev_i16 %0, 1, %1" : "=v"(d) : "v"(s));
    if constexpr (OP == 6) asm volatile("v_and_b32 %0, %1, %2" : "=v"(d) : "v"(s), "v"(0x3f803f80));
    if constexpr (OP == 7) { long long r; asm volatile("v_cvt_scalef32_pk_f32_fp8 %0, %1, %2" : "=v"(r) : "v"(s), "v"(sc)); d = (int)r; d2 = (int)(r >> 32); }
    if constexpr (OP == 8) asm volatile("v_bfi_b32 %0, %2, %1, %1" : "=v"(d) : "v"(s), "v"(0x80008000));
    if constexpr (OP == 9) asm volatile("v_pk_mul_f16 %0, %1, %1" : "=v"(d) : "v"(s));
    if constexpr (OP == 10) asm volatile("v_pk_lshlrev_b16 %0, 7, %1" : "=v"(d) : "v"(s));
    if constexpr (OP == 12) { long long r, a = ((long long)s << 32) | (unsigned)s; asm volatile("v_pk_fma_f32 %0, %1, %2, %1 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "v"(a)); d = (int)r; d2 = (int)(r >> 32); }
    if constexpr (OP == 11) asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %1, 1.0" : "=v"(d) : "v"(s));
}

// PER = vector instructions behind each MFMA (0: no MFMA at all, 64 instructions per pass)
template <int OP, int PER>
__global__ void __launch_bounds__(512) k(const int *seed, int *out, unsigned long long *cyc, int iters)
{
    int s[8], d[16], d2[16];
    for (int i = 0; i < 8; ++i) s[i] = seed[(threadIdx.x * 8 + i) & 4095];
    for (int i = 0; i < 16; ++i) { d[i] = 0; d2[i] = 0; }
    float sc = 1.0f + (float)(seed[0] & 1);
    v4f acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = v4f{0, 0, 0, 0};
    v4i fa = {s[0], s[1], s[2], s[3]}, fb = {s[4], s[5], s[6], s[7]};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if constexpr (PER == 0) {
#pragma unroll
            for (int i = 0; i < 64; ++i) one<OP>(d[i & 15], d2[i & 15], s[i & 7], sc);
        } else {
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, fb), __builtin_bit_cast(v8bf, fa), acc[m & 3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PER; ++i) one<OP>(d[(m * PER + i) & 15], d2[(m * PER + i) & 15], s[i & 7], sc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    int r = 0;
    for (int i = 0; i < 16; ++i) r += d[i] + d2[i];
    for (int i = 0; i < 4; ++i) r += (int)(acc[i].x + acc[i].y + acc[i].z + acc[i].w);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int OP, int PER>
static void run(const char *name, const int *seed, int *out, unsigned long long *cyc)
{
    const int iters = 2000;
    for (int threads : {256, 512}) {
        for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL((k<OP, PER>), dim3(256), dim3(threads), 0, 0, seed, out, cyc, iters);
        hipDeviceSynchronize();
        const int waves = 256 * threads / 64;
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double med = (double)h[waves / 2];
        if (PER == 0)
            printf("%-34s alone        waves/SIMD %d: %6.2f cycles per instruction per wave, %6.2f per SIMD\n", name, threads / 256,
                   med / (64.0 * iters), med / (64.0 * iters * (threads / 256)));
        else
            printf("%-34s %d per MFMA   waves/SIMD %d: %6.2f cycles per MFMA per SIMD (16 = the matrix pipe)\n", name, PER, threads / 256,
                   med / (16.0 * iters * (threads / 256)));
    }
}

int main()
{
    int *seed, *out; unsigned long long *cyc;
    hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    std::vector<int> h(4096);
    srand(7);
    for (int i = 0; i < 4096; ++i) {
        unsigned v = 0;
        for (int b = 0; b < 4; ++b) {
            unsigned byte = rand() & 0xFF;
            if ((byte & 0x7F) == 0x7F) byte &= 0x80;
            v |= byte << (8 * b);
        }
        h[i] = (int)v;
    }
    hipMemcpy(seed, h.data(), 4096 * 4, hipMemcpyHostToDevice);
#define ALL(OP, NAME) run<OP, 0>(NAME, seed, out, cyc); run<OP, 1>(NAME, seed, out, cyc); run<OP, 2>(NAME, seed, out, cyc); run<OP, 3>(NAME, seed, out, cyc); run<OP, 4>(NAME, seed, out, cyc);
    ALL(4, "v_fma_f32")
    ALL(12, "v_pk_fma_f32")
    ALL(0, "v_cvt_scalef32_pk_bf16_fp8 (vgpr scale)")
    ALL(11, "v_cvt_scalef32_pk_bf16_fp8 (1.0)")
    ALL(1, "v_cvt_scalef32_pk_f16_fp8")
    ALL(2, "v_cvt_pk_f32_fp8")
    ALL(7, "v_cvt_scalef32_pk_f32_fp8")
    ALL(3, "v_perm_b32")
    ALL(5, "v_pk_ashrrev_i16")
    ALL(6, "v_and_b32")
    ALL(8, "v_bfi_b32")
    ALL(9, "v_pk_mul_f16")
    ALL(10, "v_pk_lshlrev_b16")
    return 0;
}
