// Non-template helper kernels of the fp8 path (element-wise fallback, split-K combine; the odd-K padding pass is dga_rows.hip's pad_rows).  Included by
// dga_launch.hip only: they are ordinary (non-inline) kernels, one definition per library.
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

// Generic kernel: any K (also K % 16 != 0), any strides.  One thread per output
// element, fp32 running sums in the oracle's order.  Used only where the LDS-DMA
// kernel's 16-byte chunking does not apply.
__device__ __forceinline__ float e4m3fn_to_f32(uint8_t v)
{
    const uint32_t e = (v >> 3) & 15u, mnt = v & 7u;
    float r;
    if (e == 0) r = (float)mnt * 0.001953125f;  // subnormal: mnt/8 * 2^-6
    else if (e == 15u && mnt == 7u) r = __builtin_nanf("");
    else r = __uint_as_float(((e + 120u) << 23) | (mnt << 20));
    return (v & 0x80) ? -r : r;
}

__global__ void __launch_bounds__(256) gemm_fp8_blockscaled_nt_generic_kernel(const GemmParams p)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = e4m3fn_to_f32((uint8_t)threadIdx.x);
    __syncthreads();
    const int g = blockIdx.z;
    const int M = p.masked_m ? min(p.masked_m[g], p.m) : p.m;
    const int n = blockIdx.x * 16 + (threadIdx.x & 15);
    const int m = blockIdx.y * 16 + (threadIdx.x >> 4);
    if (m >= M || n >= p.n) return;
    int bg = g;
    if (p.m_indices) {  // contiguous-grouped: per-row B group
        bg = p.m_indices[m];
        if (bg < 0 || bg >= p.b_groups) return;
    }
    const int64_t mrow = p.row_index ? p.row_index[(int64_t)g * p.m + m] : (int64_t)m;   // indexed: row of the flat buffers
    const uint8_t *ar = p.a + (int64_t)g * p.a_gs + mrow * p.lda;
    const uint8_t *br = p.b + (int64_t)bg * p.b_gs + (int64_t)n * p.ldb;
    const float *sa = p.sfa + (int64_t)g * p.sfa_gs + mrow * p.sfa_ld;
    const float *sb = p.sfb + (int64_t)bg * p.sfb_gs + (int64_t)(n / 128) * p.kb_n;
    float acc = 0.f;
    for (int kb = 0; kb < p.kb_n; ++kb) {
        const int k0 = kb * 128, k1 = min(p.k, k0 + 128);
        float part = 0.f;
        for (int k = k0; k < k1; ++k) part += lut[ar[k]] * lut[br[k]];
        acc += part * (sa[kb] * sb[kb]);
    }
    const v2bf h = __builtin_convertvector(v2f{acc, 0.f}, v2bf);
    p.out[(int64_t)g * p.c_gs + mrow * p.ldc + n] = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
}

// split-K combine: out[m][n] = bf16( sum_s slab[s][m][n] ), s ascending (= k ascending; deterministic).  The fp32
// counterpart of the reference's StreamkReduceAdd (op_kernel/kernel/padding_streamk_matmul_kernel.h:96-98).
__global__ void __launch_bounds__(256) splitk_reduce_bf16_kernel(const float *partial, uint16_t *out, int64_t mn,
                                                                 int splitk)
{
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= mn) return;
    if (((mn & 7) == 0) && ((((uintptr_t)out) & 15) == 0)) {
        v4f a0 = *(const v4f *)(partial + i), a1 = *(const v4f *)(partial + i + 4);
        for (int s = 1; s < splitk; ++s) {
            a0 += *(const v4f *)(partial + (int64_t)s * mn + i);
            a1 += *(const v4f *)(partial + (int64_t)s * mn + i + 4);
        }
        const v2bf h0 = __builtin_convertvector(v2f{a0.x, a0.y}, v2bf), h1 = __builtin_convertvector(v2f{a0.z, a0.w}, v2bf);
        const v2bf h2 = __builtin_convertvector(v2f{a1.x, a1.y}, v2bf), h3 = __builtin_convertvector(v2f{a1.z, a1.w}, v2bf);
        *(v4i *)(out + i) = v4i{__builtin_bit_cast(int, h0), __builtin_bit_cast(int, h1), __builtin_bit_cast(int, h2),
                               __builtin_bit_cast(int, h3)};
    } else {
        for (int q = 0; q < 8 && i + q < mn; ++q) {
            float acc = partial[i + q];
            for (int s = 1; s < splitk; ++s) acc += partial[(int64_t)s * mn + i + q];
            const v2bf h = __builtin_convertvector(v2f{acc, 0.f}, v2bf);
            out[i + q] = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
        }
    }
}

}  // namespace dga
