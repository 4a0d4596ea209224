// Quantisers that sit immediately upstream of the fp8 GEMM in a MoE layer:
//   dga_cast_to_fp8_1x128    activations  x[rows,k] (fp32 / bf16 / fp16) -> e4m3fn bytes + fp32 scale per 1x128 block
//   dga_cast_to_fp8_128x128  weights      x[rows,k]                      -> e4m3fn bytes + fp32 scale per 128x128 block
// The reference has no fp8 data path (its inputs are fp16 files written by numpy, scripts/gen_data.py); the
// definition of record is the test oracle's quantiser (quant_1x128 / quant_128x128 in oracle/):
//   amax = max |x| over the block (NaN ignored), scale = amax / 448 (1 if amax == 0),
//   q = e4m3fn_rne_satfinite(x / scale)  with an IEEE fp32 division,
// and the results are byte-exact against it.  Both kernels are HBM streams (read 2 or 4 bytes, write 1 per element).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#include "dga_hip.h"
#include "dga_internal.hpp"

namespace dga {

typedef float v4f_c __attribute__((ext_vector_type(4)));
typedef int v4i_c __attribute__((ext_vector_type(4)));
typedef int v2i_c __attribute__((ext_vector_type(2)));

// two fp32 -> two e4m3fn bytes (low 16 bits), any input.  v_cvt_pk_fp8_f32 is OCP e4m3fn on gfx950 (round to nearest
// even, subnormals included) but turns overflow into the NaN code and every NaN into 0xFF (probed:
// scripts/ubench/probe_cvt_fp8.hip); the definition saturates (satfinite) and encodes NaN as sign | 0x7F, so inputs
// are clamped first and NaN is patched afterwards.
__device__ __forceinline__ uint32_t cvt2_e4m3fn(float a, float b)
{
    const float ca = __builtin_fminf(__builtin_fmaxf(a, -448.f), 448.f);
    const float cb = __builtin_fminf(__builtin_fmaxf(b, -448.f), 448.f);
    uint32_t r = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(ca, cb, 0, false) & 0xFFFFu;
    if (a != a) r = (r & 0xFF00u) | ((__float_as_uint(a) >> 24) & 0x80u) | 0x7Fu;
    if (b != b) r = (r & 0x00FFu) | ((((__float_as_uint(b) >> 24) & 0x80u) | 0x7Fu) << 8);
    return r;
}

// max over the 16 lanes of a DPP row (= one 1x128 block), result in every lane: row_mirror, row_half_mirror, then the
// two quad permutes -- four v_max_f32_dpp, no LDS traffic.
template <int CTRL> __device__ __forceinline__ float dpp_max(float x)
{
    const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false);
    return __builtin_fmaxf(x, __builtin_bit_cast(float, y));
}
__device__ __forceinline__ float row16_max(float x)
{
    x = dpp_max<0x140>(x);  // row_mirror: lane i <-> 15 - i
    x = dpp_max<0x141>(x);  // row_half_mirror: i <-> 7 - i inside each half
    x = dpp_max<0x4E>(x);   // quad_perm [2,3,0,1]
    x = dpp_max<0xB1>(x);   // quad_perm [1,0,3,2]
    return x;
}

// DGA_CAST_UE8M0: the block scale rounded UP to a power of two, 2^ceil(log2(amax / 448)) -- upstream DeepGEMM's use_ue8m0
// quantisation; such scales ride in the matrix instruction's E8M0 operands (DGA_POLICY_UE8M0_SCALES).  Exact on the bits: a
// scale with a non-zero mantissa moves to the next exponent.
__device__ __forceinline__ float block_scale(float amax, bool ue8m0)
{
    float s = amax > 0.f ? amax / 448.f : 1.f;
    if (ue8m0) {
        const uint32_t b = __float_as_uint(s);
        if (b & 0x007FFFFFu) s = __uint_as_float((b & 0x7F800000u) + 0x00800000u);
    }
    return s;
}

__device__ __forceinline__ bool has_ff_byte(uint32_t w) { return (((~w) - 0x01010101u) & w & 0x80808080u) != 0; }

// 8 values of one scale block -> 8 codes.  Fast path: the IEEE quotient x / s by the compiler's own fp32 division
// recurrence (rcp, one Newton step, then q, two residual corrections -- the final fma is the correctly rounded
// quotient) with the reciprocal refined once per block instead of once per element, and without the range scaling,
// which is not needed while s is far from the ends of the exponent range.  No clamp either: |x / s| <= 448 (1 + 2^-22)
// rounds to 448.  Anything unusual -- s tiny, huge, infinite, or a NaN among the inputs (the hardware's 0xFF code
// shows it) -- takes the general path: true division, clamp, NaN patch.
__device__ __forceinline__ void quant8(const float (&v)[8], float s, uint32_t &w0, uint32_t &w1)
{
    const uint32_t sb = __float_as_uint(s);
    const bool s_ok = (sb - 0x20000000u) < 0x3F000000u;  // 2^-63 <= s < 2^63
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float r1 = __builtin_fmaf(__builtin_fmaf(-s, r0, 1.f), r0, r0);
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float q0 = v[j] * r1;
        const float q1 = __builtin_fmaf(__builtin_fmaf(-s, q0, v[j]), r1, q0);
        // s > 0: the quotient has x's sign, also when it is a zero (the residual steps turn -0 into +0)
        q[j] = __builtin_copysignf(__builtin_fmaf(__builtin_fmaf(-s, q1, v[j]), r1, q1), v[j]);
    }
    w0 = ((uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false) & 0xFFFFu) |
         ((uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], 0, false) << 16);
    w1 = ((uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], 0, false) & 0xFFFFu) |
         ((uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], 0, false) << 16);
    if (!s_ok || has_ff_byte(w0) || has_ff_byte(w1)) {
        w0 = cvt2_e4m3fn(v[0] / s, v[1] / s) | (cvt2_e4m3fn(v[2] / s, v[3] / s) << 16);
        w1 = cvt2_e4m3fn(v[4] / s, v[5] / s) | (cvt2_e4m3fn(v[6] / s, v[7] / s) << 16);
    }
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int kBytes = 4;
    static __device__ __forceinline__ float load(const void *p, int64_t i) { return ((const float *)p)[i]; }
    static __device__ __forceinline__ void load8(const void *p, int64_t i, float (&v)[8])
    {
        const v4f_c lo = *(const v4f_c *)((const float *)p + i), hi = *(const v4f_c *)((const float *)p + i + 4);
        v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
    }
};
struct Bf16Tag {};
struct F16Tag {};
template <> struct Elem<Bf16Tag> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ float cv(uint32_t h) { return __uint_as_float(h << 16); }
    static __device__ __forceinline__ float load(const void *p, int64_t i) { return cv(((const uint16_t *)p)[i]); }
    static __device__ __forceinline__ void load8(const void *p, int64_t i, float (&v)[8])
    {
        const v4i_c w = *(const v4i_c *)((const uint16_t *)p + i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[2 * j] = __uint_as_float((uint32_t)w[j] << 16);
            v[2 * j + 1] = __uint_as_float((uint32_t)w[j] & 0xFFFF0000u);
        }
    }
};
template <> struct Elem<F16Tag> {
    static constexpr int kBytes = 2;
    static __device__ __forceinline__ float load(const void *p, int64_t i) { return (float)((const _Float16 *)p)[i]; }
    static __device__ __forceinline__ void load8(const void *p, int64_t i, float (&v)[8])
    {
        typedef _Float16 v8h __attribute__((ext_vector_type(8)));
        const v8h w = *(const v8h *)((const _Float16 *)p + i);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)w[j];
    }
};

// 16 lanes share one 1x128 block, 8 consecutive elements per lane: a wave covers 4 blocks per pass, loads and stores
// are contiguous per 16-lane group (256 / 512 bytes in, 128 bytes out).
template <typename T>
__global__ void __launch_bounds__(256) cast_1x128_kernel(const void *x, uint8_t *q, float *sf, int64_t rows, int64_t k,
                                                         int64_t kb_n, bool vec_in, bool vec_out, int64_t ldq, bool ue8m0)
{
    const int64_t blk = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (blk >= rows * kb_n) return;  // whole 16-lane groups leave together
    const int sub = threadIdx.x & 15;
    const int64_t row = blk / kb_n, kb = blk - row * kb_n;
    const int64_t c0 = kb * 128 + sub * 8, base = row * k + c0;
    float v[8];
    if (vec_in && c0 + 8 <= k) {
        Elem<T>::load8(x, base, v);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (c0 + j < k) ? Elem<T>::load(x, base + j) : 0.f;
    }
    float amax = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) amax = __builtin_fmaxf(amax, __builtin_fabsf(v[j]));
    amax = row16_max(amax);
    const float s = block_scale(amax, ue8m0);
    if (sub == 0) sf[blk] = s;
    uint32_t w0, w1;
    quant8(v, s, w0, w1);
    // (columns at and beyond k were read as 0 and quantise to the zero byte: they fill the row's tail up to ldq)
    const int64_t qbase = row * ldq + c0;
    if (vec_out && c0 + 8 <= ldq) {
        *(v2i_c *)(q + qbase) = v2i_c{(int)w0, (int)w1};
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < ldq) q[qbase + j] = (uint8_t)(((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xFF);
    }
}

// The same, U blocks per 16-lane group with all U loads issued before the first is used: U x 16 (32) bytes in flight per lane
// instead of one load.  Block j of a group is blk + j * stride (stride = a U-th of the blocks), so that each of the U passes of
// the grid is the contiguous stream the one-block kernel reads.  Whole rows only on the fast path (vec_in, vec_out, K % 128 == 0).
template <typename T, int U>
__global__ void __launch_bounds__(256) cast_1x128_unrolled_kernel(const void *x, uint8_t *q, float *sf, int64_t blocks, int64_t stride, bool ue8m0)
{
    const int64_t blk0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) >> 4;
    if (blk0 >= stride) return;
    const int sub = threadIdx.x & 15;
    float v[U][8];
    bool live[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
        const int64_t blk = blk0 + j * stride;
        live[j] = blk < blocks;
        if (live[j]) Elem<T>::load8(x, blk * 128 + sub * 8, v[j]);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
        if (!live[j]) continue;          // (uniform over the 16-lane group)
        const int64_t blk = blk0 + j * stride;
        float amax = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) amax = __builtin_fmaxf(amax, __builtin_fabsf(v[j][e]));
        amax = row16_max(amax);
        const float s = block_scale(amax, ue8m0);
        if (sub == 0) sf[blk] = s;
        uint32_t w0, w1;
        quant8(v[j], s, w0, w1);
        *(v2i_c *)(q + blk * 128 + sub * 8) = v2i_c{(int)w0, (int)w1};
    }
}

// one workgroup per 128x128 block: thread t holds 64 elements of row t/2 (columns 64*(t&1) ..), the block amax goes
// through LDS, nothing is read twice.
template <typename T>
__global__ void __launch_bounds__(256) cast_128x128_kernel(const void *x, uint8_t *q, float *sf, int64_t rows, int64_t k,
                                                           int64_t kb_n, bool vec_in, bool vec_out, int64_t ldq, bool ue8m0)
{
    __shared__ float red[4];
    const int64_t rb = blockIdx.x / kb_n, kb = blockIdx.x - rb * kb_n;
    const int t = threadIdx.x;
    const int64_t row = rb * 128 + (t >> 1);
    const int64_t c0 = kb * 128 + (t & 1) * 64;
    const bool row_ok = row < rows;
    float v[64];
    float amax = 0.f;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) {
        float e[8];
        const int64_t c = c0 + g8 * 8, base = row * k + c;
        if (row_ok && vec_in && c + 8 <= k) {
            Elem<T>::load8(x, base, e);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) e[j] = (row_ok && c + j < k) ? Elem<T>::load(x, base + j) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v[g8 * 8 + j] = e[j];
            amax = __builtin_fmaxf(amax, __builtin_fabsf(e[j]));
        }
    }
#pragma unroll
    for (int msk = 1; msk < 64; msk <<= 1) amax = __builtin_fmaxf(amax, __shfl_xor(amax, msk, 64));
    if ((t & 63) == 0) red[t >> 6] = amax;
    __syncthreads();
    amax = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
    const float s = block_scale(amax, ue8m0);
    if (t == 0) sf[blockIdx.x] = s;
    if (!row_ok) return;
#pragma unroll
    for (int g8 = 0; g8 < 8; ++g8) {
        const int64_t c = c0 + g8 * 8, base = row * k + c;
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) e[j] = v[g8 * 8 + j];
        uint32_t w0, w1;
        quant8(e, s, w0, w1);
        const int64_t qbase = row * ldq + c;
        if (vec_out && c + 8 <= ldq) {
            *(v2i_c *)(q + qbase) = v2i_c{(int)w0, (int)w1};
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c + j < ldq) q[qbase + j] = (uint8_t)(((j < 4 ? w0 : w1) >> (8 * (j & 3))) & 0xFF);
        }
    }
}

template <typename T>
static int launch_cast(int mode, const void *x, void *q, float *sf, int64_t rows, int64_t k, int64_t ldq, bool ue8m0, hipStream_t stream)
{
    const int64_t kb_n = (k + 127) / 128;
    // a lane's 8 elements start at element row*k + 8*j: 16-byte aligned for every row iff k % 8 == 0
    const bool vec_in = (reinterpret_cast<uintptr_t>(x) % 16 == 0) && (k % 8 == 0);
    const bool vec_out = (reinterpret_cast<uintptr_t>(q) % 8 == 0) && (ldq % 8 == 0);
    // 16-bit inputs, large problems: two blocks per 16-lane group (32 bytes in flight per lane): [32768, 7168] bf16 140.7 -> 124.2 us
    // (5.06 -> 5.73 TB/s; four blocks 129.8; fp32 inputs are faster one block at a time: 195 against 203 / 207 us; scripts/cast_ab.py)
    static const int unroll = [] { const char *e = std::getenv("DGA_CAST_UNROLL"); return e ? std::atoi(e) : 0; }();
    const bool two = unroll ? unroll >= 2 : (Elem<T>::kBytes == 2 && rows * kb_n >= 131072);
    if (mode == 0 && vec_in && vec_out && k % 128 == 0 && ldq == k && two) {
        const int64_t blocks = rows * kb_n;
        const int64_t stride = (blocks + 1) / 2;
        const int64_t grid = (stride * 16 + 255) / 256;
        if (grid > 0x7FFFFFFFll) return DGA_E_RANGE;
        if (unroll >= 4)
            hipLaunchKernelGGL((cast_1x128_unrolled_kernel<T, 4>), dim3(static_cast<unsigned>(((blocks + 3) / 4 * 16 + 255) / 256)), dim3(256), 0,
                               stream, x, static_cast<uint8_t *>(q), sf, blocks, (blocks + 3) / 4, ue8m0);
        else
            hipLaunchKernelGGL((cast_1x128_unrolled_kernel<T, 2>), dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, x,
                               static_cast<uint8_t *>(q), sf, blocks, stride, ue8m0);
        return record_hip(hipGetLastError());
    }
    if (mode == 0) {
        const int64_t blocks = rows * kb_n;
        const int64_t grid = (blocks * 16 + 255) / 256;
        if (grid > 0x7FFFFFFFll) return DGA_E_RANGE;
        hipLaunchKernelGGL(cast_1x128_kernel<T>, dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, x,
                           static_cast<uint8_t *>(q), sf, rows, k, kb_n, vec_in, vec_out, ldq, ue8m0);
    } else {
        const int64_t grid = ((rows + 127) / 128) * kb_n;
        if (grid > 0x7FFFFFFFll) return DGA_E_RANGE;
        hipLaunchKernelGGL(cast_128x128_kernel<T>, dim3(static_cast<unsigned>(grid)), dim3(256), 0, stream, x,
                           static_cast<uint8_t *>(q), sf, rows, k, kb_n, vec_in, vec_out, ldq, ue8m0);
    }
    return record_hip(hipGetLastError());
}

static int run_cast(int mode, const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, void *stream, int flags = 0)
{
    if (flags & ~DGA_CAST_UE8M0) return DGA_E_RANGE;
    const bool ue8m0 = (flags & DGA_CAST_UE8M0) != 0;
    if (rows < 0 || k < 0 || ldq < k || ldq > (k + 127) / 128 * 128) return DGA_E_SHAPE;
    if (rows == 0 || k == 0) return DGA_OK;
    if (!x || !q || !sf) return DGA_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (x_dtype) {
        case DGA_DT_FP32: return launch_cast<float>(mode, x, q, sf, rows, k, ldq, ue8m0, st);
        case DGA_DT_BF16: return launch_cast<Bf16Tag>(mode, x, q, sf, rows, k, ldq, ue8m0, st);
        case DGA_DT_FP16: return launch_cast<F16Tag>(mode, x, q, sf, rows, k, ldq, ue8m0, st);
        default: return DGA_E_DTYPE;
    }
}

}  // namespace dga

extern "C" {

int dga_cast_to_fp8_1x128(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, float *sf, void *stream)
{
    return dga::run_cast(0, x, x_dtype, rows, k, q, k, sf, stream);
}

int dga_cast_to_fp8_128x128(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, float *sf, void *stream)
{
    return dga::run_cast(1, x, x_dtype, rows, k, q, k, sf, stream);
}

int dga_cast_to_fp8_1x128_ld(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, void *stream)
{
    return dga::run_cast(0, x, x_dtype, rows, k, q, ldq, sf, stream);
}

int dga_cast_to_fp8_128x128_ld(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, void *stream)
{
    return dga::run_cast(1, x, x_dtype, rows, k, q, ldq, sf, stream);
}

int dga_cast_to_fp8_1x128_ex(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, int flags, void *stream)
{
    return dga::run_cast(0, x, x_dtype, rows, k, q, ldq, sf, stream, flags);
}

int dga_cast_to_fp8_128x128_ex(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, int flags, void *stream)
{
    return dga::run_cast(1, x, x_dtype, rows, k, q, ldq, sf, stream, flags);
}

}  // extern "C"
