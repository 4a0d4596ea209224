// How each gfx950 MFMA form that can carry e4m3 data sums one 128-wide k block, against the oracle's order
// (fp32, k ascending).  Development aid for the "strict" dispatch policy; prints a table, asserts nothing.
//   F0  v_mfma_scale_f32_16x16x128_f8f6f4  (the fast path's instruction; unit E8M0 scales)
//   F1  v_mfma_f32_16x16x32_fp8_fp8 x 4    (C chained)
//   F2  v_mfma_f32_16x16x32_bf16 x 4       (e4m3 -> bf16 is exact; C chained)
//   F3  v_mfma_f32_16x16x4_f32 x 32        (e4m3 -> f32 exact; C chained: claimed = a k-ordered fmaf chain)
//   F4  v_mfma_f32_32x32x2_f32 x 64        (same claim, 32x32 tile)
// Build: hipcc --offload-arch=gfx950 -O2 -o probe_mfma_forms probe_mfma_forms.hip   (make -C scripts/ubench)
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
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

static float e4m3_to_f32(uint8_t v)
{
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float r;
    if (e == 15 && m == 7) return NAN;
    r = e == 0 ? ldexpf((float)m, -9) : ldexpf((float)(8 + m), e - 10);
    return s ? -r : r;
}
static uint8_t f32_to_e4m3(float x)  // RNE, satfinite
{
    uint8_t best = 0;
    float bd = INFINITY;
    const uint8_t sign = std::signbit(x) ? 0x80 : 0;
    const float ax = fabsf(x);
    for (int c = 0; c < 0x7F; ++c) {
        const float d = fabsf(e4m3_to_f32((uint8_t)c) - ax);
        if (d < bd || (d == bd && (c & 1) == 0)) { bd = d; best = (uint8_t)c; }
    }
    return sign | best;
}

// A, B: [rows][128] e4m3 bytes (rows = 16, or 32 for F4); D[i][j] = sum_k A[i][k] B[j][k]; out: [rows][rows] fp32
__global__ void probe_kernel(const uint8_t *A, const uint8_t *B, float *out, int form)
{
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    if (form == 0) {
        v8i a, b;
        const int *pa = (const int *)(A + r * 128), *pb = (const int *)(B + r * 128);
        for (int j = 0; j < 4; ++j) {
            a[j] = pa[4 * q + j]; a[4 + j] = pa[16 + 4 * q + j];
            b[j] = pb[4 * q + j]; b[4 + j] = pb[16 + 4 * q + j];
        }
        // D rows come from the first operand, columns from the second
        v4f d = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, v4f{0, 0, 0, 0}, 0, 0, 0, 0, 0, 0);
        for (int t = 0; t < 4; ++t) out[(4 * q + t) * 16 + r] = d[t];
    } else if (form == 1) {
        v4f d = {0, 0, 0, 0};
        for (int s = 0; s < 4; ++s) {
            const long a = *(const long *)(A + r * 128 + 32 * s + 8 * q);
            const long b = *(const long *)(B + r * 128 + 32 * s + 8 * q);
            d = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a, b, d, 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) out[(4 * q + t) * 16 + r] = d[t];
    } else if (form == 2) {
        v4f d = {0, 0, 0, 0};
        for (int s = 0; s < 4; ++s) {
            v8bf a, b;
            for (int j = 0; j < 8; ++j) {
                const int ka = 32 * s + 8 * q + j;
                const float fa = __builtin_amdgcn_cvt_f32_fp8((int)A[r * 128 + ka], 0);
                const float fb = __builtin_amdgcn_cvt_f32_fp8((int)B[r * 128 + ka], 0);
                a[j] = (__bf16)fa; b[j] = (__bf16)fb;
            }
            d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d, 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) out[(4 * q + t) * 16 + r] = d[t];
    } else if (form == 3) {
        v4f d = {0, 0, 0, 0};
        for (int s = 0; s < 32; ++s) {
            const float fa = __builtin_amdgcn_cvt_f32_fp8((int)A[r * 128 + 4 * s + q], 0);
            const float fb = __builtin_amdgcn_cvt_f32_fp8((int)B[r * 128 + 4 * s + q], 0);
            d = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, d, 0, 0, 0);
        }
        for (int t = 0; t < 4; ++t) out[(4 * q + t) * 16 + r] = d[t];
    } else {
        const int r32 = l & 31, h = l >> 5;
        v16f d;
        for (int t = 0; t < 16; ++t) d[t] = 0.f;
        for (int s = 0; s < 64; ++s) {
            const float fa = __builtin_amdgcn_cvt_f32_fp8((int)A[r32 * 128 + 2 * s + h], 0);
            const float fb = __builtin_amdgcn_cvt_f32_fp8((int)B[r32 * 128 + 2 * s + h], 0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, d, 0, 0, 0);
        }
        for (int t = 0; t < 16; ++t) out[((t & 3) + 8 * (t >> 2) + 4 * h) * 32 + r32] = d[t];
    }
}

struct Stat { double max_rel_exact = 0, sum_rel_exact = 0, max_rel_seq = 0; long n = 0, bit_equal = 0; };

int main()
{
    const int TRIALS = 400;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    uint8_t *dA, *dB; float *dO;
    hipMalloc(&dA, 32 * 128); hipMalloc(&dB, 32 * 128); hipMalloc(&dO, 32 * 32 * 4);
    const char *names[5] = {"F0 f8f6f4 16x16x128", "F1 fp8_fp8 16x16x32 x4", "F2 bf16 16x16x32 x4", "F3 f32 16x16x4 x32", "F4 f32 32x32x2 x64"};
    for (int dataset = 0; dataset < 2; ++dataset) {
        Stat st[5];
        for (int t = 0; t < TRIALS; ++t) {
            std::vector<uint8_t> A(32 * 128), B(32 * 128);
            for (int i = 0; i < 32; ++i) {
                if (dataset == 0) {  // amax-quantised normal data (bench.py's recipe)
                    float xa[128], xb[128], ma = 0, mb = 0;
                    for (int k = 0; k < 128; ++k) { xa[k] = nd(rng); xb[k] = nd(rng); ma = fmaxf(ma, fabsf(xa[k])); mb = fmaxf(mb, fabsf(xb[k])); }
                    for (int k = 0; k < 128; ++k) { A[i * 128 + k] = f32_to_e4m3(xa[k] * 448.f / ma); B[i * 128 + k] = f32_to_e4m3(xb[k] * 448.f / mb); }
                } else {             // arbitrary finite bit patterns
                    for (int k = 0; k < 128; ++k) {
                        uint8_t a = rng() & 0xFF, b = rng() & 0xFF;
                        if ((a & 0x7F) == 0x7F) a ^= 1;
                        if ((b & 0x7F) == 0x7F) b ^= 1;
                        A[i * 128 + k] = a; B[i * 128 + k] = b;
                    }
                }
            }
            hipMemcpy(dA, A.data(), 32 * 128, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), 32 * 128, hipMemcpyHostToDevice);
            for (int f = 0; f < 5; ++f) {
                const int R = f == 4 ? 32 : 16;
                std::vector<float> got(R * R);
                hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO, f);
                hipMemcpy(got.data(), dO, R * R * 4, hipMemcpyDeviceToHost);
                for (int i = 0; i < R; ++i)
                    for (int j = 0; j < R; ++j) {
                        double ex = 0, S = 0;
                        volatile float seq = 0.f;
                        for (int k = 0; k < 128; ++k) {
                            const float pa = e4m3_to_f32(A[i * 128 + k]), pb = e4m3_to_f32(B[j * 128 + k]);
                            const float pr = pa * pb;  // exact
                            ex += (double)pr; S += fabs((double)pr);
                            seq = seq + pr;
                        }
                        const float g = got[i * R + j];
                        const double re = fabs((double)g - ex) / S, rs = fabs((double)g - (double)seq) / S;
                        st[f].max_rel_exact = fmax(st[f].max_rel_exact, re);
                        st[f].sum_rel_exact += re;
                        st[f].max_rel_seq = fmax(st[f].max_rel_seq, rs);
                        st[f].n++;
                        uint32_t ug, us; const float sq = seq; memcpy(&ug, &g, 4); memcpy(&us, &sq, 4);
                        st[f].bit_equal += (ug == us);
                    }
            }
        }
        printf("dataset %s (K = 128, %d tiles)\n", dataset == 0 ? "amax-quantised normal" : "arbitrary finite bytes", TRIALS);
        for (int f = 0; f < 5; ++f)
            printf("  %-26s max|got-exact|/S = 2^%6.2f   mean = 2^%6.2f   max|got-seq32|/S = 2^%6.2f   bit-equal to seq32: %ld / %ld\n",
                   names[f], log2(st[f].max_rel_exact + 1e-300), log2(st[f].sum_rel_exact / st[f].n + 1e-300),
                   log2(st[f].max_rel_seq + 1e-300), st[f].bit_equal, st[f].n);
    }
    // cancellation: a = [X, -X, v x 126], b = [X, X, w x 126]: exact = 126 v w
    printf("cancellation a=[X,-X,v..] b=[X,X,w..]: exact 126*v*w\n");
    const float Xs[3] = {448.f, 64.f, 1.f};
    const float vs[4] = {1.f, 0.125f, 0.015625f, 0.001953125f};
    for (float X : Xs)
        for (float v : vs) {
            std::vector<uint8_t> A(32 * 128, 0), B(32 * 128, 0);
            A[0] = f32_to_e4m3(X); A[1] = f32_to_e4m3(-X); B[0] = B[1] = f32_to_e4m3(X);
            for (int k = 2; k < 128; ++k) { A[k] = f32_to_e4m3(v); B[k] = f32_to_e4m3(v); }
            hipMemcpy(dA, A.data(), 32 * 128, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), 32 * 128, hipMemcpyHostToDevice);
            printf("  X=%5g v=w=%-10g exact=%-12g", X, v, 126.0 * v * v);
            for (int f = 0; f < 5; ++f) {
                float g = 0;
                hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dO, f);
                hipMemcpy(&g, dO, 4, hipMemcpyDeviceToHost);
                printf(" F%d=%-12g", f, g);
            }
            printf("\n");
        }
    return 0;
}
