// Does the scaled fp8 matrix instruction keep fp32 when it ACCUMULATES (C-in), and what does a whole K chain cost in accuracy
// when the block scales ride in its E8M0 operands instead of a vector-pipe promotion?  (VERDICT r4 item 3, step A.)
//   part 1  deterministic C-in probes: is a small product lost against a large C, and against which alignment?
//   part 2  K = 4096 / 7168 chains of 128-wide blocks on the bench recipe with power-of-two (UE8M0) scales:
//             HW   acc = v_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, e8m0(sfa), e8m0(sfb))        (no vector work at all)
//             SW   part = mfma(a, b, 0); acc = fma(part, sfa*sfb, acc)                             (the fast policy today)
//             H32  the 32x32x64 form, two per block, scales in its operands
//           each against the oracle's value (per-block fp32 k-ascending sum, acc += part * (sfa*sfb)), rounded to bf16
//   part 3  rate: back-to-back scaled MFMAs with live scale registers, 16 independent accumulators, 1 and 2 waves per SIMD
// Development aid; prints tables, asserts nothing.  make -C scripts/ubench probe_mfma_scale_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

static float e4m3_to_f32(uint8_t v)
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    const float r = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
    return s ? -r : r;
}
static uint8_t f32_to_e4m3(float x)   // RNE, saturating, |x| <= 448 expected
{
    const uint8_t sign = std::signbit(x) ? 0x80 : 0;
    float ax = fabsf(x);
    if (!(ax > 0.f)) return sign;
    if (ax >= 448.f) return sign | 0x7E;
    int e;
    frexpf(ax, &e);                       // ax = f * 2^e, f in [0.5, 1)
    int E = e - 1;                        // ax = 1.xxx * 2^E
    if (E < -6) E = -6;                   // subnormal grid: step 2^-9
    const float step = ldexpf(1.f, E - 3);
    const float q = nearbyintf(ax / step);   // RNE (default rounding mode)
    const float v = q * step;
    // encode v
    if (v < ldexpf(1.f, -6)) return sign | (uint8_t)(int)q;           // subnormal: m = q
    int e2; frexpf(v, &e2); const int EE = e2 - 1;
    const int m = (int)(v / ldexpf(1.f, EE - 3)) - 8;
    return sign | (uint8_t)(((EE + 7) << 3) | m);
}
static uint16_t bf16_rne(float f)
{
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static int bf16_key(uint16_t b) { const int mag = b & 0x7FFF; return (b & 0x8000) ? -mag : mag; }

// lane's 32 bytes of a row of 128 (the fast kernel's fragment order: bytes [16q,16q+16) and [64+16q, 64+16q+16))
__device__ inline v8i frag16(const uint8_t *row, int q)
{
    const int *p = (const int *)row;
    v8i v;
    for (int j = 0; j < 4; ++j) { v[j] = p[4 * q + j]; v[4 + j] = p[16 + 4 * q + j]; }
    return v;
}

// part 1: one MFMA with a caller-given C
__global__ void cin_kernel(const uint8_t *A, const uint8_t *B, const float *C, float *out, int ea, int eb)
{
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    const v8i a = frag16(A + r * 128, q), b = frag16(B + r * 128, q);
    v4f c;
    for (int t = 0; t < 4; ++t) c[t] = C[(4 * q + t) * 16 + r];
    const int sa = ea * 0x01010101, sb = eb * 0x01010101;
    const v4f d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int t = 0; t < 4; ++t) out[(4 * q + t) * 16 + r] = d[t];
}

// part 2: A [nb][16][128], B [nb][16][128], sfa [nb][16] (power of two), sfb [nb] (power of two); out [3][16][16] (+ 32x32 unused)
// D[i][j]: i = row of the FIRST operand.  We feed B rows first (the product kernel's orientation): D[n][m].
__global__ void chain_kernel(const uint8_t *A, const uint8_t *B, const float *sfa, const float *sfb, float *out, int nb)
{
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    v4f hw = {0, 0, 0, 0}, sw = {0, 0, 0, 0};
    for (int kb = 0; kb < nb; ++kb) {
        const v8i a = frag16(A + (size_t)(kb * 16 + r) * 128, q), b = frag16(B + (size_t)(kb * 16 + r) * 128, q);
        const float sa = sfa[kb * 16 + r], sb = sfb[kb];          // this lane's A row is r (second operand: column of D)
        const int ea = (__float_as_int(sa) >> 23) & 0xFF, eb = (__float_as_int(sb) >> 23) & 0xFF;
        // first operand = B rows (scale sfb, uniform), second operand = A rows (scale sfa of row r)
        hw = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, hw, 0, 0, 0, eb * 0x01010101, 0, ea * 0x01010101);
        const v4f part = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b, a, v4f{0, 0, 0, 0}, 0, 0, 0, 0, 0, 0);
        const float s = sa * sb;
        for (int t = 0; t < 4; ++t) sw[t] = __builtin_fmaf(part[t], s, sw[t]);
    }
    // d[t] = D[n = 4q + t][m = r]  -> out[m][n]
    for (int t = 0; t < 4; ++t) { out[r * 16 + 4 * q + t] = hw[t]; out[256 + r * 16 + 4 * q + t] = sw[t]; }
}

// the 32x32x64 form: A, B rows 0..31; lane (r32 = l & 31, h = l >> 5) holds 32 bytes of row r32: k = 32h .. 32h+31 of each 64-half
__global__ void chain32_kernel(const uint8_t *A, const uint8_t *B, const float *sfa, const float *sfb, float *out, int nb)
{
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    v16f hw;
    for (int t = 0; t < 16; ++t) hw[t] = 0.f;
    for (int kb = 0; kb < nb; ++kb) {
        const float sa = sfa[kb * 32 + r], sb = sfb[kb];
        const int ea = (__float_as_int(sa) >> 23) & 0xFF, eb = (__float_as_int(sb) >> 23) & 0xFF;
        for (int half = 0; half < 2; ++half) {
            const int *pa = (const int *)(A + (size_t)(kb * 32 + r) * 128 + 64 * half + 32 * h);
            const int *pb = (const int *)(B + (size_t)(kb * 32 + r) * 128 + 64 * half + 32 * h);
            v8i a, b;
            for (int j = 0; j < 8; ++j) { a[j] = pa[j]; b[j] = pb[j]; }
            hw = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, hw, 0, 0, 0, eb * 0x01010101, 0, ea * 0x01010101);
        }
    }
    // d[t]: row (of the first operand, n) = (t & 3) + 8 (t >> 2) + 4 h, column m = r
    for (int t = 0; t < 16; ++t) out[r * 32 + (t & 3) + 8 * (t >> 2) + 4 * h] = hw[t];
}

// part 3: rate
template <int MODE>   // 0: scaled 16x16x128 accumulate, scales from registers; 1: unscaled form (constant 0 scales); 2: scaled 32x32x64
__global__ void __launch_bounds__(512) rate_kernel(const int *seed, float *out, int iters)
{
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = seed[(threadIdx.x * 8 + i) & 4095]; b[i] = seed[(threadIdx.x * 8 + i + 2048) & 4095]; }
    int sa = 0x7F7F7F7F - (seed[threadIdx.x & 4095] & 0x01010101), sb = 0x7F7F7F7F - (seed[(threadIdx.x + 77) & 4095] & 0x01010101);
    float r = 0;
    if (MODE == 2) {
        v16f acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb);
            asm volatile("" : "+v"(sa), "+v"(sb));
        }
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) r += acc[i][j];
    } else {
        v4f acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = v4f{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = MODE == 0 ? __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, sa, 0, sb)
                                   : __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0, 0, 0);
            asm volatile("" : "+v"(sa), "+v"(sb));
        }
        for (int i = 0; i < 16; ++i) r += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

struct Stat {
    double max_rel = 0, sum_rel = 0; long n = 0, gt2 = 0, gt1 = 0; int max_ulp = 0;
    void add(float got, float oracle, double S)
    {
        const double re = fabs((double)got - (double)oracle) / S;
        max_rel = fmax(max_rel, re); sum_rel += re; ++n;
        const int u = abs(bf16_key(bf16_rne(got)) - bf16_key(bf16_rne(oracle)));
        if (u > 2) ++gt2;
        if (u > 1) ++gt1;
        if (u > max_ulp) max_ulp = u;
    }
    void print(const char *name) const
    {
        printf("  %-34s max|d|/S = 2^%6.2f  mean = 2^%6.2f   bf16: > 1 ulp %.3e  > 2 ulp %.3e  max %d ulp   (%ld outputs)\n", name,
               log2(max_rel + 1e-300), log2(sum_rel / (double)n + 1e-300), (double)gt1 / (double)n, (double)gt2 / (double)n, max_ulp, n);
    }
};

int main(int argc, char **argv)
{
    const int TRIALS = argc > 1 ? atoi(argv[1]) : 300;
    uint8_t *dA, *dB; float *dC, *dO, *dsa, *dsb;
    const int NBMAX = 56;
    hipMalloc(&dA, NBMAX * 32 * 128); hipMalloc(&dB, NBMAX * 32 * 128); hipMalloc(&dC, 1024 * 4); hipMalloc(&dO, 4096 * 4);
    hipMalloc(&dsa, NBMAX * 32 * 4); hipMalloc(&dsb, NBMAX * 4);

    // ---------------------------------------------------------------- part 1
    printf("part 1: C-in of v_mfma_scale_f32_16x16x128_f8f6f4 (one product p = a*b in the block, everything else zero, C uniform)\n");
    {
        struct Case { float c, a, b; int ea, eb; } cases[] = {
            {8388608.f, 1.f, 1.f, 127, 127},          // 2^23 + 1: representable; lost if C is aligned to fewer than 24 bits
            {4194304.f, 1.f, 1.f, 127, 127},          // 2^22 + 1
            {1048576.f, 1.f, 1.f, 127, 127},          // 2^20 + 1
            {65536.f, 1.f, 1.f, 127, 127},            // 2^16 + 1
            {8192.f, 1.f, 1.f, 127, 127},             // 2^13 + 1
            {1.f, 0.001953125f, 0.001953125f, 127, 127},   // 1 + 2^-18
            {1.f, 0.001953125f, 0.015625f, 127, 127},      // 1 + 2^-15
            {1.f, 0.015625f, 0.015625f, 127, 127},         // 1 + 2^-12
            {1.f, 1.f, 1.f, 127 - 20, 127},                // 1 + 2^-20 through the scale
            {1.f, 1.f, 1.f, 127 - 12, 127 - 11},           // 1 + 2^-23 through both scales
            {-448.f * 448.f, 448.f, 448.f, 127, 127},      // exact cancellation against C
            {3.f, 1.5f, 1.5f, 127 + 1, 127},               // 3 + 4.5
            {1e-30f, 1.f, 1.f, 127, 127},
        };
        for (const Case &cs : cases) {
            std::vector<uint8_t> A(16 * 128, 0), B(16 * 128, 0);
            std::vector<float> C(256, cs.c), O(256);
            for (int r = 0; r < 16; ++r) { A[r * 128 + 37] = f32_to_e4m3(cs.a); B[r * 128 + 37] = f32_to_e4m3(cs.b); }
            hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(cin_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO, cs.ea, cs.eb);
            hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
            const double want = (double)cs.c + (double)cs.a * cs.b * ldexp(1.0, cs.ea - 127 + cs.eb - 127);
            const float want32 = (float)want;
            printf("  C = %-14.9g p = %-12.6g scale 2^%-4d  want %-18.10g (fp32 %-16.10g)  got %-16.10g  %s\n", cs.c, cs.a * cs.b, cs.ea + cs.eb - 254, want,
                   want32, O[5 * 16 + 3], O[5 * 16 + 3] == want32 ? "= fp32(C + p)" : "DIFFERS");
        }
        // many small products against a large C: C = 2^20, 128 products of 1*1 -> 2^20 + 128; and of 2^-3 * 1 -> 2^20 + 16
        for (float v : {1.f, 0.125f, 0.015625f}) {
            std::vector<uint8_t> A(16 * 128, f32_to_e4m3(v)), B(16 * 128, f32_to_e4m3(1.f));
            std::vector<float> C(256, 1048576.f), O(256);
            hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
            hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(cin_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO, 127, 127);
            hipMemcpy(O.data(), dO, 1024, hipMemcpyDeviceToHost);
            printf("  C = 2^20, 128 products of %g: want %.10g got %.10g\n", v, 1048576.0 + 128.0 * v, O[0]);
        }
    }

    // ---------------------------------------------------------------- part 2
    std::mt19937 rng(4321);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int nb : {32, 56}) {
        Stat hw, sw, h32, hw_vs_sw;
        Stat hwE, swE;   // against the exact value (fp64), not the oracle's fp32 chain
        for (int t = 0; t < TRIALS; ++t) {
            std::vector<uint8_t> A(nb * 32 * 128), B(nb * 32 * 128);
            std::vector<float> sfa(nb * 32), sfb(nb);
            for (int kb = 0; kb < nb; ++kb) {
                // B: one 128x128 scale block; only 32 of its rows are used, the other 96 only feed the amax
                static float xb[128][128];
                float mb = 0;
                for (int i = 0; i < 128; ++i) for (int k = 0; k < 128; ++k) { xb[i][k] = nd(rng); mb = fmaxf(mb, fabsf(xb[i][k])); }
                const float sb = ldexpf(1.f, (int)ceilf(log2f(mb / 448.f)));
                sfb[kb] = sb;
                for (int i = 0; i < 32; ++i) for (int k = 0; k < 128; ++k) B[(size_t)(kb * 32 + i) * 128 + k] = f32_to_e4m3(xb[i][k] / sb);
                for (int i = 0; i < 32; ++i) {
                    float xa[128], ma = 0;
                    for (int k = 0; k < 128; ++k) { xa[k] = nd(rng); ma = fmaxf(ma, fabsf(xa[k])); }
                    const float sa = ldexpf(1.f, (int)ceilf(log2f(ma / 448.f)));
                    sfa[kb * 32 + i] = sa;
                    for (int k = 0; k < 128; ++k) A[(size_t)(kb * 32 + i) * 128 + k] = f32_to_e4m3(xa[k] / sa);
                }
            }
            // 16-row view for the 16x16 kernels: rows 0..15 of each block, repacked
            std::vector<uint8_t> A16(nb * 16 * 128), B16(nb * 16 * 128); std::vector<float> sfa16(nb * 16);
            for (int kb = 0; kb < nb; ++kb) for (int i = 0; i < 16; ++i) {
                memcpy(&A16[(size_t)(kb * 16 + i) * 128], &A[(size_t)(kb * 32 + i) * 128], 128);
                memcpy(&B16[(size_t)(kb * 16 + i) * 128], &B[(size_t)(kb * 32 + i) * 128], 128);
                sfa16[kb * 16 + i] = sfa[kb * 32 + i];
            }
            std::vector<float> o16(512), o32(1024);
            hipMemcpy(dA, A16.data(), A16.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B16.data(), B16.size(), hipMemcpyHostToDevice);
            hipMemcpy(dsa, sfa16.data(), sfa16.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dsb, sfb.data(), sfb.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(chain_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dO, nb);
            hipMemcpy(o16.data(), dO, 2048, hipMemcpyDeviceToHost);
            hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
            hipMemcpy(dsa, sfa.data(), sfa.size() * 4, hipMemcpyHostToDevice);
            hipLaunchKernelGGL(chain32_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dO, nb);
            hipMemcpy(o32.data(), dO, 4096, hipMemcpyDeviceToHost);
            for (int m = 0; m < 32; ++m)
                for (int n = 0; n < 32; ++n) {
                    double ex = 0, S = 0;
                    float acc = 0.f;
                    for (int kb = 0; kb < nb; ++kb) {
                        volatile float part = 0.f;
                        double pe = 0, ps = 0;
                        for (int k = 0; k < 128; ++k) {
                            const float pr = e4m3_to_f32(A[(size_t)(kb * 32 + m) * 128 + k]) * e4m3_to_f32(B[(size_t)(kb * 32 + n) * 128 + k]);
                            part = part + pr; pe += pr; ps += fabs((double)pr);
                        }
                        const float s = sfa[kb * 32 + m] * sfb[kb];
                        volatile float prod = part * s;
                        acc = acc + prod;
                        ex += pe * (double)s; S += ps * (double)s;
                    }
                    h32.add(o32[m * 32 + n], acc, S);
                    if (m < 16 && n < 16) {
                        hw.add(o16[m * 16 + n], acc, S); sw.add(o16[256 + m * 16 + n], acc, S);
                        hw_vs_sw.add(o16[m * 16 + n], o16[256 + m * 16 + n], S);
                        hwE.add(o16[m * 16 + n], (float)ex, S); swE.add(o16[256 + m * 16 + n], (float)ex, S);
                    }
                }
        }
        printf("part 2: K = %d (%d blocks), bench recipe with power-of-two scales, %d trials; reference = the oracle's fp32 chain\n", nb * 128, nb, TRIALS);
        hw.print("HW scale, C-in accumulate 16x16x128");
        sw.print("SW promotion (fast policy today)");
        h32.print("HW scale, C-in accumulate 32x32x64");
        hw_vs_sw.print("HW against SW");
        hwE.print("HW against fp64-exact");
        swE.print("SW against fp64-exact");
    }

    // ---------------------------------------------------------------- part 3
    {
        const int iters = 2000;
        int *seed; float *out;
        hipMalloc(&seed, 4096 * 4); hipMalloc(&out, 256 * 512 * 4);
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
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = argc > 2 ? atoi(argv[2]) : 200;
        printf("part 3: rate (256 workgroups, random e4m3 operands, %d launches per row, the last one reported)\n", reps);
        for (int mode = 0; mode < 3; ++mode)
            for (int threads : {256, 512}) {
                float ms = 0;
                for (int rep = 0; rep < reps; ++rep) {
                    hipEventRecord(e0);
                    if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                    else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                    else hipLaunchKernelGGL(rate_kernel<2>, dim3(256), dim3(threads), 0, 0, seed, out, iters);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                const double flops = 2.0 * 16 * 16 * 128 * 16.0 * iters * (threads / 64) * 256;
                printf("  %-44s waves/SIMD %d: %.3f ms  %.0f TFLOP/s\n",
                       mode == 0 ? "scaled 16x16x128, C-in, scales in registers" : mode == 1 ? "unscaled 16x16x128, C-in" : "scaled 32x32x64, C-in, scales in registers",
                       threads / 256, ms, flops / ms / 1e9);
            }
    }
    return 0;
}
