// One-launch split-K inside a workgroup for the decode rows of the 16-bit operator (catlass_dynamic_matmul, NT, bf16 / fp16 in and
// out): the 16-bit form of gemm_fp8_wskc_kernel (gemm_fp8_wsk_kernel.hpp).  The 8 waves of a workgroup take the 8 K slices of the
// workgroup's n-tiles (16 columns each, all M <= 16 or 32 rows); every wave stages its k steps (64 elements = 128 bytes per row) through
// a private ring of D stages in LDS by LDS-DMA -- whole 128-byte lines, the tile kernels' XOR image -- and reads fragments back with
// ds_read_b128; no barrier while streaming, hand-counted vmcnt (every vector-memory instruction of the loop is inline-asm DMA); the
// ring runs on across the passes of a workgroup that walks more n-tiles than one pass holds; the eight fp32 partial tiles meet in a
// slab behind the rings and are summed in slice order, then rounded to the operator's 16-bit type.  No slab in HBM, no combine launch.
//
// Same K slices (ceil(KS / 8) k steps each), same per-slice arithmetic (two v_mfma_f32_16x16x32 per k step, chunk kg then chunk
// 4 + kg, accumulated in k order) and the same combine order as the tile kernel's two-launch split-K with factor 8
// (gemm_b16_kernel.hpp + splitk_reduce_16_kernel): bit-identical to it (tests/test_op16_wsk_gpu.py).
// Reference counterparts: the operator's device entry (aclnn_catlass_dynamic_matmul/op_kernel/catlass_dynamic_matmul.cpp:16-45), the
// Stream-K kernel's fused reduce (op_kernel/kernel/padding_streamk_matmul_kernel.h:92-107) and the single-core split-K kernel types
// of op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36.
#pragma once
#include "gemm_b16_kernel.hpp"

namespace dga {

template <bool BF16, int TN, int D, int TM = 1>   // TM: 16-row tiles of x (M <= 16 TM)
__global__ void __launch_bounds__(512) gemm_b16_wsk_kernel(const B16Params p)
{
    constexpr int WAVES = 8, BM = 16 * TM, BNW = TN * 16, ROWS = BM + BNW, NT = WAVES * 64;
    constexpr int L = ROWS / 8;                                  // DMA instructions per stage: 8 rows each
    constexpr int STAGE = ROWS * 128, RING = D * STAGE, SLAB = BM * BNW * 4;
    static_assert(WAVES * (RING + SLAB) <= 160 * 1024, "LDS of one CU");
    static_assert((D - 1) * L < 64, "vmcnt");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, kg = lane >> 4;
    constexpr uint32_t kOutOfRange = 0x80000000u;

    const int M = p.m, KS = p.k / 64;                            // k steps (p.k is a multiple of 64)
    const int nt_total = (p.n + 15) >> 4, G = gridDim.x;
    const int nt0 = (int)(((int64_t)blockIdx.x * nt_total) / G), nt1 = (int)(((int64_t)(blockIdx.x + 1) * nt_total) / G);
    if (nt1 <= nt0) return;
    const int npass = (nt1 - nt0 + TN - 1) / TN;
    const int ksps = p.ks_per_split;                            // the host's slice length: ceil(KS / 8), see launch_b16_wsk
    const int c0 = wave * ksps, c1 = min(KS, c0 + ksps);
    const int len = max(0, c1 - c0);                            // k steps of this wave's slice
    const int s_eff = (KS + ksps - 1) / ksps;                   // waves that own at least one k step

    uint8_t *ring = smem + wave * RING;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(lptr_t)smem + wave * RING;
    float *slab = (float *)(smem + WAVES * RING);               // [wave][m][n] fp32
    const int d_row = lane >> 3;
    const v4i a_rsrc = make_rsrc(p.x, (int64_t)M * p.ldx * 2);
    uint32_t a_voff[BM / 8];
    int col[2];   // (the swizzle looks at the row inside its 16-row tile: instruction j's column depends on j & 1 only)
#pragma unroll
    for (int j = 0; j < 2; ++j) col[j] = ((lane & 7) ^ swz_a((8 * j + d_row) & 15)) * 16;
#pragma unroll
    for (int j = 0; j < BM / 8; ++j) {
        const int row = 8 * j + d_row;
        a_voff[j] = row < M ? (uint32_t)row * (uint32_t)(p.ldx * 2) + col[j & 1] : kOutOfRange;   // rows at or beyond M: zero-filled
    }
    const int f_off0 = li * 128 + ((kg ^ swz_a(li)) * 16), f_off1 = li * 128 + (((4 + kg) ^ swz_a(li)) * 16);

    // ---- the refill cursor: pass ip, k step ik of the slice, ring stage istg; D positions ahead of the multiplication
    int ip = 0, ik = 0, istg = 0;
    v4i b_rsrc = a_rsrc;
    uint32_t b_voff[BNW / 8];
    auto set_issue_pass = [&](int pass) {
        const int ntc = nt0 + pass * TN, cnt = min(TN, nt1 - ntc), n0 = ntc * 16;
        b_rsrc = make_rsrc(p.yt + (int64_t)n0 * p.ldy, (int64_t)(p.n - n0) * p.ldy * 2);
#pragma unroll
        for (int j = 0; j < BNW / 8; ++j) {
            const int row = 8 * j + d_row;
            b_voff[j] = (row < cnt * 16 && n0 + row < p.n) ? (uint32_t)row * (uint32_t)(p.ldy * 2) + col[j & 1] : kOutOfRange;
        }
    };
    // the next stage of the sequence (past its end: every lane out of range -- zeros land, nothing is fetched -- so that the
    // number of instructions in flight stays what the waits assume)
    auto issue_next = [&]() {
        const uint32_t base = ring_lds + istg * STAGE;
        const bool live = ip < npass;
        const int k0 = (c0 + ik) * 128;   // bytes
#pragma unroll
        for (int j = 0; j < ROWS / 8; ++j) {
            uint32_t vo = j < BM / 8 ? a_voff[j < BM / 8 ? j : 0] : b_voff[j >= BM / 8 ? j - BM / 8 : 0];
            vo = live ? vo : kOutOfRange;
            dma16(vo, j < BM / 8 ? a_rsrc : b_rsrc, (uint32_t)k0, base + j * 1024);
        }
        istg = istg + 1 == D ? 0 : istg + 1;
        if (live && ++ik == len) {
            ik = 0;
            if (++ip < npass) set_issue_pass(ip);
        }
    };
    set_issue_pass(0);
    if (len > 0) {
#pragma unroll
        for (int d = 0; d < D; ++d) issue_next();
    }

    int cstg = 0;
    for (int pass = 0; pass < npass; ++pass) {
        const int ntc = nt0 + pass * TN, cnt = min(TN, nt1 - ntc), n0 = ntc * 16;
        v4f acc[TM][TN];
#pragma unroll
        for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[u][j] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < len; ++i) {
            wait_vmcnt<(D - 1) * L>();   // the oldest stage has landed; the D - 1 younger ones stay in flight
            const uint8_t *st = ring + cstg * STAGE;
            v4i a0[TM], a1[TM], b0[TN], b1[TN];
#pragma unroll
            for (int u = 0; u < TM; ++u) {
                a0[u] = *(const v4i *)(st + u * 2048 + f_off0);
                a1[u] = *(const v4i *)(st + u * 2048 + f_off1);
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                b0[t] = *(const v4i *)(st + (TM + t) * 2048 + f_off0);
                b1[t] = *(const v4i *)(st + (TM + t) * 2048 + f_off1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the stage is in registers: refill it
            issue_next();
            cstg = cstg + 1 == D ? 0 : cstg + 1;
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                if (t < cnt) {
#pragma unroll
                    for (int u = 0; u < TM; ++u) {
                        acc[u][t] = mfma_b16<BF16>(b0[t], a0[u], acc[u][t]);
                        acc[u][t] = mfma_b16<BF16>(b1[t], a1[u], acc[u][t]);
                    }
                }
            }
        }
        // ---- the pass's partial tiles meet in the slab behind the rings (the rings keep streaming the next pass): lane (li, kg)
        //      owns rows m = 16 u + li, columns 16 t + 4 kg + [0, 4)
#pragma unroll
        for (int u = 0; u < TM; ++u)
#pragma unroll
            for (int t = 0; t < TN; ++t) *(v4f *)(slab + (size_t)wave * BM * BNW + (16 * u + li) * BNW + t * 16 + 4 * kg) = acc[u][t];
        __syncthreads();
        for (int g = tid; g < BM * BNW; g += NT) {
            const int m = g / BNW, nl = g % BNW;
            if (m >= M || nl >= cnt * 16 || n0 + nl >= p.n) continue;
            float a = slab[m * BNW + nl];
            for (int s = 1; s < s_eff; ++s) a += slab[(size_t)s * BM * BNW + m * BNW + nl];   // s ascending = k ascending
            uint16_t *dst = p.z16 + (int64_t)m * p.n + n0 + nl;
            if constexpr (BF16) {
                const v2bf h = __builtin_convertvector(v2f{a, 0.f}, v2bf);
                *dst = (uint16_t)(__builtin_bit_cast(uint32_t, h) & 0xFFFFu);
            } else {
                *dst = __builtin_bit_cast(uint16_t, (_Float16)a);
            }
        }
        if (pass + 1 < npass) __syncthreads();   // everyone has read the slab before the next pass's partial tiles go there
    }
    wait_vmcnt<0>();   // the refills past the sequence (zeros) have landed
}

}  // namespace dga
