// Strict dispatch policy (dispatchPolicyTag = DGA_POLICY_STRICT): the block-scaled fp8 NT GEMM computed in the
// reference CPU path's own arithmetic -- fp32 products, fp32 running sum, k ascending
// (/root/reference/deep_gemm_ascend/framework/tests/test.py:37, generate_code.hpp:216,320-335: fp32 L0C accumulator) --
// so that its output is bit-identical to the definition of record:
//     partial = sum_{k in block, ascending} f32(a) * f32(b)        (each product is exact in fp32: 4 x 4 significand bits)
//     acc     = acc + partial * (sfa * sfb)                        (fp32 multiply, then fp32 add: no fma)
//
// The fast path's v_mfma_scale_f32_16x16x128_f8f6f4 (and v_mfma_f32_16x16x32_fp8_fp8, same datapath) aligns every octet
// of products to its largest exponent and drops what falls ~13 bits below it (scripts/ubench/probe_mfma_forms.hip,
// profiles/r02_mfma_forms.txt); the bf16 form keeps fp32-like sums but in its own order.  The f32-input MFMA
// v_mfma_f32_16x16x4_f32 is a k-ordered chain of fp32 fma on its accumulator -- with exact products that IS the
// oracle's sequential sum, measured bit-equal on 102 400 / 102 400 random sums.  32 chained instructions per 128-wide k
// block; lane (r = lane & 15, q = lane >> 4) feeds k = 4*step + q, so the chain runs k = 0, 1, 2, ... in order.
// Rate: the fp32 matrix rate (157 TFLOP/s peak), 1/32 of the fast path's -- this is the opt-in exact form, not the
// throughput form.  It takes every shape (any K, any alignment), dense, masked-grouped and contiguous-grouped.
#pragma once
#include "gemm_fp8_kernel.hpp"

namespace dga {

// acc + partial * s with two roundings (the oracle is built with -ffp-contract=off)
__device__ __forceinline__ float promote_no_fma(float acc, float partial, float s)
{
#pragma clang fp contract(off)
    const float scaled = partial * s;
    return acc + scaled;
}
__device__ __forceinline__ float mul_no_fma(float a, float b)
{
#pragma clang fp contract(off)
    return a * b;
}

// 4 x 4 byte transpose of a 16-byte chunk: dword q of the result holds bytes q, 4 + q, 8 + q, 12 + q of the chunk -- the four k
// values lane group q of the 16x16x4 MFMA feeds over a chunk's four steps, in step order.  Eight v_perm_b32 per chunk at staging
// time buy a conversion without a per-lane shift, two steps per v_cvt_pk_f32_fp8: next to fp32 MFMAs every vector instruction is
// paid in matrix-pipe time (scripts/ubench/f32x_ubench: the pipe alone 0.99 of its peak, with a shift + a conversion per operand
// value 0.68).
__device__ __forceinline__ v4i transpose_bytes_4x4(v4i w)
{
    const uint32_t a = __builtin_amdgcn_perm((uint32_t)w.y, (uint32_t)w.x, 0x05010400u), b = __builtin_amdgcn_perm((uint32_t)w.y, (uint32_t)w.x, 0x07030602u);
    const uint32_t c = __builtin_amdgcn_perm((uint32_t)w.w, (uint32_t)w.z, 0x05010400u), d = __builtin_amdgcn_perm((uint32_t)w.w, (uint32_t)w.z, 0x07030602u);
    return v4i{(int)__builtin_amdgcn_perm(c, a, 0x05040100u), (int)__builtin_amdgcn_perm(c, a, 0x07060302u),
               (int)__builtin_amdgcn_perm(d, b, 0x05040100u), (int)__builtin_amdgcn_perm(d, b, 0x07060302u)};
}

// TM = m-tiles (16 rows) per wave; workgroup = 2 x 2 waves, tile (32*TM) x 128, one LDS stage, register prefetch.
// The launcher uses TM = 2 (two workgroups per CU cover each other's barriers: 110 TFLOP/s at 4096^3 against 91 for TM = 4)
// and TM = 1 where 64-row tiles would leave CUs idle.
template <int TM>
__global__ void __launch_bounds__(256) gemm_fp8_strict_nt_kernel(const GemmParams p)
{
    constexpr int BM = 32 * TM, BN = 128, TN = 4;
    constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;  // 16-byte chunks per thread per k block
    __shared__ __attribute__((aligned(16))) uint8_t smem[(BM + BN) * 128 + (BM + 4) * 4];
    uint8_t *lds_a = smem, *lds_b = smem + BM * 128;
    float *lds_s = (float *)(smem + (BM + BN) * 128);  // [0,BM) sfa rows, [BM] sfb

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 15, q = lane >> 4;

    const int tiles_per_group = p.tiles_m * p.tiles_n;
    const int g = blockIdx.x / tiles_per_group;
    const int t_in = blockIdx.x - g * tiles_per_group;
    const int tm = t_in % p.tiles_m, tn = t_in / p.tiles_m;
    const int M = p.masked_m ? min(p.masked_m[g], p.m) : p.m;
    const int m0 = tm * BM, n0 = tn * BN;
    if (m0 >= M) return;
    int bg = g;
    if (p.m_indices) {  // contiguous-grouped: BM divides the segment alignment, so a tile lies in one segment
        bg = p.m_indices[m0];
        if (bg < 0 || bg >= p.b_groups) return;
    }
    const int64_t *ridx = p.row_index ? p.row_index + (int64_t)g * p.m : nullptr;   // slot -> row of the flat buffers
    const uint8_t *A = p.a + (int64_t)g * p.a_gs;
    const uint8_t *B = p.b + (int64_t)bg * p.b_gs;
    const float *SFA = p.sfa + (int64_t)g * p.sfa_gs;
    const float *SFB = p.sfb + (int64_t)bg * p.sfb_gs + (int64_t)(n0 / 128) * p.kb_n;
    uint16_t *C = p.out + (int64_t)g * p.c_gs;
    // 16-byte vector loads need K % 16 == 0 and 16-byte aligned operands; otherwise bytes are gathered one by one
    const bool vec = ((p.k & 15) == 0) && ((p.lda & 15) == 0) && ((p.ldb & 15) == 0) &&
                     ((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0;

    v4i ra[A_CH], rb[B_CH];
    float rs = 0.f;
    auto fetch_chunk = [&](const uint8_t *row, int kc) -> v4i {
        if (vec) {
            if (kc + 16 <= p.k) return *(const v4i *)(row + kc);
            return v4i{0, 0, 0, 0};
        }
        uint32_t w[4] = {0, 0, 0, 0};
        for (int j = 0; j < 16; ++j)
            if (kc + j < p.k) w[j >> 2] |= (uint32_t)row[kc + j] << (8 * (j & 3));
        return v4i{(int)w[0], (int)w[1], (int)w[2], (int)w[3]};
    };
    // row pointers of this thread's chunks (row clamp, row table, chunk column): once per tile, not once per k block
    const uint8_t *a_ptr[A_CH], *b_ptr[B_CH];
    int a_col[A_CH], b_col[B_CH];
#pragma unroll
    for (int it = 0; it < A_CH; ++it) {
        const int c = it * 256 + tid, row = c >> 3;
        const int mr = m0 + min(row, M - 1 - m0);
        a_col[it] = (c & 7) * 16;
        a_ptr[it] = A + (ridx ? ridx[mr] : (int64_t)mr) * p.lda;
    }
#pragma unroll
    for (int it = 0; it < B_CH; ++it) {
        const int c = it * 256 + tid, row = c >> 3;
        b_col[it] = (c & 7) * 16;
        b_ptr[it] = B + (int64_t)(n0 + min(row, p.n - 1 - n0)) * p.ldb;
    }
    const float *s_ptr = nullptr;
    if (tid < BM) {
        const int mr = min(m0 + tid, M - 1);
        s_ptr = SFA + (ridx ? ridx[mr] : (int64_t)mr) * p.sfa_ld;
    } else if (tid == BM) {
        s_ptr = SFB;   // BM <= 128 < 256 threads
    }
    auto fetch = [&](int kb) {
        const int k0 = kb * 128;
#pragma unroll
        for (int it = 0; it < A_CH; ++it) ra[it] = fetch_chunk(a_ptr[it], k0 + a_col[it]);
#pragma unroll
        for (int it = 0; it < B_CH; ++it) rb[it] = fetch_chunk(b_ptr[it], k0 + b_col[it]);
        if (s_ptr) rs = s_ptr[kb];
    };
    auto stage = [&]() {
#pragma unroll
        for (int it = 0; it < A_CH; ++it) {
            const int c = it * 256 + tid, row = c >> 3, ch = c & 7;
            *(v4i *)(lds_a + row * 128 + ((ch ^ swz_a(row)) * 16)) = transpose_bytes_4x4(ra[it]);
        }
#pragma unroll
        for (int it = 0; it < B_CH; ++it) {
            const int c = it * 256 + tid, row = c >> 3, ch = c & 7;
            *(v4i *)(lds_b + row * 128 + ((ch ^ swz_a(row)) * 16)) = transpose_bytes_4x4(rb[it]);
        }
        if (tid <= BM) lds_s[tid] = rs;
    };

    v4f acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = v4f{0.f, 0.f, 0.f, 0.f};

    const int a_row = wm * (16 * TM) + r;   // + 16*mt
    const int b_row = wn * 64 + r;          // + 16*nt

    if (p.kb_n > 0) fetch(0);
    for (int kb = 0; kb < p.kb_n; ++kb) {
        __syncthreads();   // every wave has left the previous k block's LDS image
        stage();
        __syncthreads();
        if (kb + 1 < p.kb_n) fetch(kb + 1);   // lands under this block's MFMAs

        v4f part[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) part[i][j] = v4f{0.f, 0.f, 0.f, 0.f};
        // chunk c + 1's dwords are read while chunk c's 32 MFMAs are issued (the loop is unrolled: the two register sets alternate)
        int ca[2][TM], cb[2][TN];      // this lane's dword of a chunk: k = 16c + q, + 4, + 8, + 12
        auto read_chunk = [&](int c, int (&da)[TM], int (&db)[TN]) {
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int row = a_row + 16 * mt;
                da[mt] = *(const int *)(lds_a + row * 128 + ((c ^ swz_a(row)) * 16) + 4 * q);
            }
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                const int row = b_row + 16 * nt;
                db[nt] = *(const int *)(lds_b + row * 128 + ((c ^ swz_a(row)) * 16) + 4 * q);
            }
        };
        read_chunk(0, ca[0], cb[0]);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if (c + 1 < 8) read_chunk(c + 1, ca[(c + 1) & 1], cb[(c + 1) & 1]);
            // the chunk's twelve packed conversions first (steps 4c .. 4c + 3: two steps per conversion), then its 32 MFMAs: a
            // conversion whose result the very next MFMA reads stalls that MFMA behind the vector write (1103 -> 1050 us at 4096^3;
            // a whole chunk ahead instead costs 20 more registers, the third wave per SIMD with them, and measures 1078)
            v2f fa[2][TM], fb[2][TN];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) fa[h][mt] = h ? __builtin_amdgcn_cvt_pk_f32_fp8(ca[c & 1][mt], true) : __builtin_amdgcn_cvt_pk_f32_fp8(ca[c & 1][mt], false);
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) fb[h][nt] = h ? __builtin_amdgcn_cvt_pk_f32_fp8(cb[c & 1][nt], true) : __builtin_amdgcn_cvt_pk_f32_fp8(cb[c & 1][nt], false);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt)
                            part[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(d ? fb[h][nt].y : fb[h][nt].x, d ? fa[h][mt].y : fa[h][mt].x, part[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // two-level dequant in the oracle's order: s = sfa * sfb (rounded), acc = acc + partial * s (two roundings)
        const float sfb_v = lds_s[BM];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const float s = mul_no_fma(lds_s[a_row + 16 * mt], sfb_v);
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) {
                // (scalar on purpose: v_pk_mul_f32 + v_pk_add_f32 here measured 1181 against 1130 us at 4096^3)
                acc[mt][nt].x = promote_no_fma(acc[mt][nt].x, part[mt][nt].x, s);
                acc[mt][nt].y = promote_no_fma(acc[mt][nt].y, part[mt][nt].y, s);
                acc[mt][nt].z = promote_no_fma(acc[mt][nt].z, part[mt][nt].z, s);
                acc[mt][nt].w = promote_no_fma(acc[mt][nt].w, part[mt][nt].w, s);
            }
        }
    }

    // epilogue: D[i][j] with i = n (4q + t), j = m (r): a lane owns 4 consecutive n of one row
    const bool vec_st = ((p.ldc & 3) == 0) && ((((uintptr_t)C) & 7) == 0);
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int m = m0 + a_row + 16 * mt;
        if (m >= M) continue;
        if (p.m_indices && p.m_indices[m] != bg) continue;
        uint16_t *crow = C + (ridx ? ridx[m] : (int64_t)m) * p.ldc;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            const int n = n0 + wn * 64 + 16 * nt + 4 * q;
            const v4f v = acc[mt][nt];
            const v2bf h0 = __builtin_convertvector(v2f{v.x, v.y}, v2bf);
            const v2bf h1 = __builtin_convertvector(v2f{v.z, v.w}, v2bf);
            const int w0 = __builtin_bit_cast(int, h0), w1 = __builtin_bit_cast(int, h1);
            if (vec_st && n + 4 <= p.n) {
                *(int2 *)(crow + n) = int2{w0, w1};
            } else {
                const uint16_t e[4] = {(uint16_t)(w0 & 0xFFFF), (uint16_t)((uint32_t)w0 >> 16), (uint16_t)(w1 & 0xFFFF),
                                       (uint16_t)((uint32_t)w1 >> 16)};
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    if (n + t < p.n) crow[n + t] = e[t];
            }
        }
    }
}

}  // namespace dga
