/*
 * dga_oracle.c -- CPU restatement of the hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (deepgemm_ascend_amd + libdga_hip.so) never does.
 *
 * What is restated, and from where (paths relative to /root/reference):
 *
 *  - "reference CPU path": golden = np.matmul(x1.astype(f32), x2.astype(f32)).astype(f32)
 *      deep_gemm_ascend/framework/tests/test.py:37
 *      deep_gemm_ascend/framework/benchmark/benchmark.py:362
 *      deep_gemm_ascend/scripts/gen_golden.py:14-15
 *    -> dga_oracle_matmul_f32_{nn,nt}()  (fp32 products, fp32 running sum, k ascending)
 *
 *  - device accumulation: fp32 accumulator in L0C, fixpipe to GM
 *      deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:216,320-335,349-364
 *    -> the same fp32 running sum.
 *
 *  - verifier: np.isclose(rtol, atol=1e-9, equal_nan=True), pass iff mismatch ratio <= 1e-4
 *      deep_gemm_ascend/scripts/verify.py:14-35, framework/benchmark/benchmark.py:384-398
 *    -> dga_oracle_verify_isclose()
 *
 *  PARITY UNPINNED (no reference code or test exists):
 *  OCP e4m3fn decode, per-1x128 / per-128x128 block scaling, bf16 rounding,
 *  grouped masked-M semantics.  The definition of record is this file:
 *     acc = 0 (fp32)
 *     for each 128-wide k block kb (ascending):
 *         partial = sum_{k in block, ascending} f32(a[m,k]) * f32(b[n,k])   (fp32 adds)
 *         acc     = acc + partial * (sfa[m,kb] * sfb[n/128,kb])             (fp32, no fma)
 *     out[m,n] = bf16_rne(acc)
 *  With unit scales and K <= 128 this is exactly the reference golden formula
 *  above followed by a bf16 rounding; with unit scales and K > 128 it differs
 *  from it only by fp32 summation order (compared in bf16 ULPs).
 *  Masked grouped semantics follow upstream DeepGEMM's public convention:
 *  rows >= masked_m[g] of out[g] are left untouched.
 *
 * Build: see oracle/Makefile (-O2 -ffp-contract=off so that no fma is formed).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#define DGA_ORACLE_API __attribute__((visibility("default")))

/* ---- OCP FP8 E4M3FN: 1 sign, 4 exponent (bias 7), 3 mantissa; no inf;
 *      S.1111.111 is NaN; S.0000.mmm subnormal = mmm/8 * 2^-6.  max = 448. ---- */
DGA_ORACLE_API float dga_oracle_e4m3fn_to_f32(uint8_t v)
{
    const int sign = v >> 7;
    const int exp = (v >> 3) & 0xF;
    const int man = v & 0x7;
    float r;
    if (exp == 0xF && man == 0x7) {
        return sign ? -NAN : NAN;
    }
    if (exp == 0) {
        r = ldexpf((float)man, -9); /* man/8 * 2^-6 */
    } else {
        r = ldexpf((float)(8 + man), exp - 7 - 3);
    }
    return sign ? -r : r;
}

DGA_ORACLE_API void dga_oracle_e4m3fn_table(float out[256])
{
    for (int i = 0; i < 256; ++i) out[i] = dga_oracle_e4m3fn_to_f32((uint8_t)i);
}

/* fp32 -> OCP e4m3fn, round-to-nearest-even, saturating to +-448 (the usual
 * "satfinite" cast used by quantisers); NaN -> 0x7F|sign. */
DGA_ORACLE_API uint8_t dga_oracle_f32_to_e4m3fn(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint8_t sign = (uint8_t)((u >> 24) & 0x80);
    if (isnan(x)) return sign | 0x7F;
    float ax = fabsf(x);
    if (ax >= 464.0f) return sign | 0x7E;  /* > midpoint(448, 480) saturates; 464 ties-to-even -> 480 -> sat */
    if (ax < 0.0009765625f) {              /* < 2^-10 = half of min subnormal 2^-9 */
        return sign;                       /* rounds to zero (tie at 2^-10 goes to even = 0) */
    }
    int e;
    float fr = frexpf(ax, &e);             /* ax = fr * 2^e, fr in [0.5,1) */
    int exp = e - 1;                       /* ax = (2fr) * 2^exp, 2fr in [1,2) */
    if (exp < -6) {
        /* subnormal: value = q * 2^-9, q in 0..7 (8 -> min normal) */
        float q = ax * 512.0f;
        float rq = nearbyintf(q);          /* default rounding mode: RNE */
        int iq = (int)rq;
        if (iq >= 8) return sign | 0x08;
        return sign | (uint8_t)iq;
    }
    float m = (fr * 2.0f - 1.0f) * 8.0f;   /* mantissa in [0,8) */
    int im = (int)nearbyintf(m);
    if (im == 8) { im = 0; exp += 1; }
    int be = exp + 7;
    if (be > 15 || (be == 15 && im == 7)) return sign | 0x7E;
    return sign | (uint8_t)((be << 3) | im);
}

/* fp32 -> bf16 round-to-nearest-even; NaN stays NaN (quiet). */
DGA_ORACLE_API uint16_t dga_oracle_f32_to_bf16(float x)
{
    uint32_t u;
    memcpy(&u, &x, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) {
        return (uint16_t)((u >> 16) | 0x0040u);
    }
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

DGA_ORACLE_API float dga_oracle_bf16_to_f32(uint16_t h)
{
    uint32_t u = ((uint32_t)h) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

/* ---- the reference CPU path restated (test.py:37 / benchmark.py:362) ---- */
/* C[M,N] = A[M,K] . B[K,N], all fp32, k ascending, fp32 running sum. */
DGA_ORACLE_API void dga_oracle_matmul_f32_nn(const float *a, const float *b, float *c,
                                             int64_t m, int64_t n, int64_t k)
{
    for (int64_t i = 0; i < m; ++i) {
        for (int64_t j = 0; j < n; ++j) {
            float acc = 0.0f;
            for (int64_t p = 0; p < k; ++p) acc += a[i * k + p] * b[p * n + j];
            c[i * n + j] = acc;
        }
    }
}

/* NT form: B is physical [N,K] (the aclnn op's convention,
 * aclnn_catlass_dynamic_matmul/op_host/catlass_dynamic_matmul_tiling.cpp:83-84). */
DGA_ORACLE_API void dga_oracle_matmul_f32_nt(const float *a, const float *b, float *c,
                                             int64_t m, int64_t n, int64_t k)
{
    for (int64_t i = 0; i < m; ++i) {
        for (int64_t j = 0; j < n; ++j) {
            float acc = 0.0f;
            const float *ar = a + i * k, *br = b + j * k;
            for (int64_t p = 0; p < k; ++p) acc += ar[p] * br[p];
            c[i * n + j] = acc;
        }
    }
}

/* ---- block-scaled fp8 NT GEMM, definition of record (see header) ---- */
static float g_tab[256];
static int g_tab_ready = 0;
static void ensure_tab(void)
{
    if (!g_tab_ready) {
        dga_oracle_e4m3fn_table(g_tab);
        g_tab_ready = 1;
    }
}

/* rows [row_begin,row_end) only: lets the caller thread over rows. */
DGA_ORACLE_API int dga_oracle_gemm_fp8_fp8_bf16_nt_rows(
    const uint8_t *a, const float *sfa, const uint8_t *b, const float *sfb, uint16_t *out,
    int64_t m, int64_t n, int64_t k, int64_t row_begin, int64_t row_end, float *out_f32 /*nullable*/)
{
    if (m < 0 || n < 0 || k < 0) return -1;
    ensure_tab();
    const int64_t kb_n = (k + 127) / 128;
    for (int64_t i = row_begin; i < row_end && i < m; ++i) {
        const uint8_t *ar = a + i * k;
        for (int64_t j = 0; j < n; ++j) {
            const uint8_t *br = b + j * k;
            float acc = 0.0f;
            for (int64_t kb = 0; kb < kb_n; ++kb) {
                const int64_t k0 = kb * 128;
                const int64_t k1 = (k0 + 128 < k) ? k0 + 128 : k;
                float partial = 0.0f;
                for (int64_t p = k0; p < k1; ++p) partial += g_tab[ar[p]] * g_tab[br[p]];
                const float s = sfa[i * kb_n + kb] * sfb[(j / 128) * kb_n + kb];
                const float scaled = partial * s;
                acc = acc + scaled;
            }
            out[i * n + j] = dga_oracle_f32_to_bf16(acc);
            if (out_f32) out_f32[i * n + j] = acc;
        }
    }
    return 0;
}

DGA_ORACLE_API int dga_oracle_gemm_fp8_fp8_bf16_nt(
    const uint8_t *a, const float *sfa, const uint8_t *b, const float *sfb, uint16_t *out,
    int64_t m, int64_t n, int64_t k)
{
    return dga_oracle_gemm_fp8_fp8_bf16_nt_rows(a, sfa, b, sfb, out, m, n, k, 0, m, NULL);
}

/* fp64 tie-break reference: exact products, fp64 sums, scales applied in fp64. */
DGA_ORACLE_API int dga_oracle_gemm_fp8_fp8_f64_nt(
    const uint8_t *a, const float *sfa, const uint8_t *b, const float *sfb, double *out,
    int64_t m, int64_t n, int64_t k)
{
    if (m < 0 || n < 0 || k < 0) return -1;
    ensure_tab();
    const int64_t kb_n = (k + 127) / 128;
    for (int64_t i = 0; i < m; ++i) {
        for (int64_t j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int64_t kb = 0; kb < kb_n; ++kb) {
                const int64_t k0 = kb * 128;
                const int64_t k1 = (k0 + 128 < k) ? k0 + 128 : k;
                double partial = 0.0;
                for (int64_t p = k0; p < k1; ++p)
                    partial += (double)g_tab[a[i * k + p]] * (double)g_tab[b[j * k + p]];
                acc += partial * ((double)sfa[i * kb_n + kb] * (double)sfb[(j / 128) * kb_n + kb]);
            }
            out[i * n + j] = acc;
        }
    }
    return 0;
}

/* grouped, masked-M: a [G,Mmax,K], sfa [G,Mmax,KB], b [G,N,K], sfb [G,NB,KB],
 * out [G,Mmax,N]; rows >= masked_m[g] of out[g] are NOT written. */
DGA_ORACLE_API int dga_oracle_m_grouped_gemm_fp8_fp8_bf16_nt_masked(
    const uint8_t *a, const float *sfa, const uint8_t *b, const float *sfb, uint16_t *out,
    const int32_t *masked_m, int64_t groups, int64_t m_max, int64_t n, int64_t k,
    int64_t group_begin, int64_t group_end)
{
    if (groups < 0 || m_max < 0 || n < 0 || k < 0) return -1;
    const int64_t kb_n = (k + 127) / 128;
    const int64_t nb_n = (n + 127) / 128;
    for (int64_t g = group_begin; g < group_end && g < groups; ++g) {
        int64_t mm = masked_m[g];
        if (mm < 0 || mm > m_max) return -2;
        int rc = dga_oracle_gemm_fp8_fp8_bf16_nt_rows(
            a + g * m_max * k, sfa + g * m_max * kb_n, b + g * n * k, sfb + g * nb_n * kb_n,
            out + g * m_max * n, mm, n, k, 0, mm, NULL);
        if (rc) return rc;
    }
    return 0;
}

/* ---- quantiser used to make synthetic inputs (per-1x128 rows / per-128x128 blocks):
 *      scale = amax/448 (1.0 if amax==0), q = e4m3fn(x/scale).  Input-generation
 *      helper, not part of the GEMM definition. ---- */
/* ue8m0 != 0: the scale rounded UP to a power of two, 2^ceil(log2(amax / 448)) -- upstream DeepGEMM's use_ue8m0 quantisation
 * (ceil_to_ue8m0: 2^ceil(log2 x)); stated on the bits so that there is no libm in it: a scale with a non-zero mantissa moves to
 * the next exponent.  No reference counterpart (the reference has no quantiser at all). */
static float dga_oracle_block_scale(float amax, int ue8m0)
{
    float s = amax > 0.0f ? amax / 448.0f : 1.0f;
    if (ue8m0) {
        uint32_t b;
        memcpy(&b, &s, 4);
        if (b & 0x007FFFFFu) { b = (b & 0x7F800000u) + 0x00800000u; memcpy(&s, &b, 4); }
    }
    return s;
}

DGA_ORACLE_API void dga_oracle_quant_1x128_ex(const float *x, uint8_t *q, float *sf, int64_t rows, int64_t k, int ue8m0)
{
    const int64_t kb_n = (k + 127) / 128;
    for (int64_t i = 0; i < rows; ++i) {
        for (int64_t kb = 0; kb < kb_n; ++kb) {
            const int64_t k0 = kb * 128, k1 = (k0 + 128 < k) ? k0 + 128 : k;
            float amax = 0.0f;
            for (int64_t p = k0; p < k1; ++p) amax = fmaxf(amax, fabsf(x[i * k + p]));
            const float s = dga_oracle_block_scale(amax, ue8m0);
            sf[i * kb_n + kb] = s;
            for (int64_t p = k0; p < k1; ++p) q[i * k + p] = dga_oracle_f32_to_e4m3fn(x[i * k + p] / s);
        }
    }
}

DGA_ORACLE_API void dga_oracle_quant_128x128_ex(const float *x, uint8_t *q, float *sf, int64_t rows, int64_t k, int ue8m0)
{
    const int64_t kb_n = (k + 127) / 128, rb_n = (rows + 127) / 128;
    for (int64_t rb = 0; rb < rb_n; ++rb) {
        const int64_t r0 = rb * 128, r1 = (r0 + 128 < rows) ? r0 + 128 : rows;
        for (int64_t kb = 0; kb < kb_n; ++kb) {
            const int64_t k0 = kb * 128, k1 = (k0 + 128 < k) ? k0 + 128 : k;
            float amax = 0.0f;
            for (int64_t i = r0; i < r1; ++i)
                for (int64_t p = k0; p < k1; ++p) amax = fmaxf(amax, fabsf(x[i * k + p]));
            const float s = dga_oracle_block_scale(amax, ue8m0);
            sf[rb * kb_n + kb] = s;
            for (int64_t i = r0; i < r1; ++i)
                for (int64_t p = k0; p < k1; ++p) q[i * k + p] = dga_oracle_f32_to_e4m3fn(x[i * k + p] / s);
        }
    }
}

DGA_ORACLE_API void dga_oracle_quant_1x128(const float *x, uint8_t *q, float *sf, int64_t rows, int64_t k)
{
    const int64_t kb_n = (k + 127) / 128;
    for (int64_t i = 0; i < rows; ++i) {
        for (int64_t kb = 0; kb < kb_n; ++kb) {
            const int64_t k0 = kb * 128, k1 = (k0 + 128 < k) ? k0 + 128 : k;
            float amax = 0.0f;
            for (int64_t p = k0; p < k1; ++p) amax = fmaxf(amax, fabsf(x[i * k + p]));
            const float s = amax > 0.0f ? amax / 448.0f : 1.0f;
            sf[i * kb_n + kb] = s;
            for (int64_t p = k0; p < k1; ++p) q[i * k + p] = dga_oracle_f32_to_e4m3fn(x[i * k + p] / s);
        }
    }
}

DGA_ORACLE_API void dga_oracle_quant_128x128(const float *x, uint8_t *q, float *sf, int64_t rows, int64_t k)
{
    const int64_t kb_n = (k + 127) / 128, rb_n = (rows + 127) / 128;
    for (int64_t rb = 0; rb < rb_n; ++rb) {
        const int64_t r0 = rb * 128, r1 = (r0 + 128 < rows) ? r0 + 128 : rows;
        for (int64_t kb = 0; kb < kb_n; ++kb) {
            const int64_t k0 = kb * 128, k1 = (k0 + 128 < k) ? k0 + 128 : k;
            float amax = 0.0f;
            for (int64_t i = r0; i < r1; ++i)
                for (int64_t p = k0; p < k1; ++p) amax = fmaxf(amax, fabsf(x[i * k + p]));
            const float s = amax > 0.0f ? amax / 448.0f : 1.0f;
            sf[rb * kb_n + kb] = s;
            for (int64_t i = r0; i < r1; ++i)
                for (int64_t p = k0; p < k1; ++p) q[i * k + p] = dga_oracle_f32_to_e4m3fn(x[i * k + p] / s);
        }
    }
}

/* ---- verifier restated (verify.py:14-35, benchmark.py:384-398).
 * Returns mismatch count; *ratio = mismatches / size.  The two reference bugs
 * under numpy 2 (size mismatch broadcast, divide by zero on empty input) are
 * defined here: size mismatch is the caller's error (-1 from the Python wrapper);
 * empty input => ratio 0, pass. ---- */
DGA_ORACLE_API int64_t dga_oracle_verify_isclose(const float *output, const float *golden, int64_t size,
                                                 double rtol, double atol, double *ratio)
{
    int64_t bad = 0;
    for (int64_t i = 0; i < size; ++i) {
        const float o = output[i], g = golden[i];
        int close;
        if (isnan(o) || isnan(g)) {
            close = isnan(o) && isnan(g); /* equal_nan=True */
        } else if (isinf(o) || isinf(g)) {
            close = (o == g);
        } else {
            close = fabs((double)o - (double)g) <= atol + rtol * fabs((double)g);
        }
        bad += !close;
    }
    if (ratio) *ratio = size > 0 ? (double)bad / (double)size : 0.0;
    return bad;
}
