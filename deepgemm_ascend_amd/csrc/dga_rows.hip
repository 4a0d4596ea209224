// Indexed row copy: the pack / unpack step on either side of the expert-sharding all-to-all
// (deepgemm_ascend_amd/parallel.py).  No reference counterpart: the reference has no collective and no
// token routing (SURVEY.md section 2 "Parallelism / communication: none implemented"); this belongs to row 8(e).
//   dst[ (dst_index ? dst_index[r] : r) * dst_stride + 0 .. row_bytes ) = src[ (src_index ? src_index[r] : r) * src_stride + ... )
// HBM-bound byte work: one workgroup per row (several for long rows), 16-byte lanes when everything is 16-byte aligned.
#include <hip/hip_runtime.h>
#include <cstdint>

#include "dga_hip.h"
#include "dga_internal.hpp"

namespace dga {

typedef int v4i __attribute__((ext_vector_type(4)));

template <bool VEC>
__global__ void __launch_bounds__(256) copy_rows_kernel(uint8_t *dst, int64_t dst_stride, const int64_t *dst_index,
                                                        const uint8_t *src, int64_t src_stride, const int64_t *src_index,
                                                        int64_t row_bytes, int64_t rows, int parts)
{
    const int64_t r = blockIdx.x / parts;
    const int part = blockIdx.x % parts;
    if (r >= rows) return;
    const int64_t dr = dst_index ? dst_index[r] : r, sr = src_index ? src_index[r] : r;
    if (dr < 0 || sr < 0) return;  // a negative index marks a row that takes no part (dga_route_tokens: id out of range)
    uint8_t *d = dst + dr * dst_stride;
    const uint8_t *s = src + sr * src_stride;
    if (VEC) {
        const int64_t chunks = row_bytes / 16;
        for (int64_t c = part * 256 + threadIdx.x; c < chunks; c += 256 * parts)
            *(v4i *)(d + c * 16) = *(const v4i *)(s + c * 16);
    } else {
        for (int64_t c = part * 256 + threadIdx.x; c < row_bytes; c += 256 * parts) d[c] = s[c];
    }
}

// Two row streams with shared indices in one launch (the dispatch packs K fp8 bytes and K/128 fp32 scales of a token into
// one payload row, and unpacks them again on the receiving side): stream 1 rides in the same workgroups.
__global__ void __launch_bounds__(256) copy_rows2_kernel(uint8_t *d0, int64_t d0_stride, const uint8_t *s0, int64_t s0_stride,
                                                         int64_t bytes0, uint8_t *d1, int64_t d1_stride, const uint8_t *s1,
                                                         int64_t s1_stride, int64_t bytes1, const int64_t *dst_index,
                                                         const int64_t *src_index, int64_t rows, int parts, int vec)
{
    const int64_t r = blockIdx.x / parts;
    const int part = blockIdx.x % parts;
    if (r >= rows) return;
    const int64_t dr = dst_index ? dst_index[r] : r, sr = src_index ? src_index[r] : r;
    if (dr < 0 || sr < 0) return;
    auto run = [&](uint8_t *d, const uint8_t *s, int64_t bytes) {
        if (vec) {
            for (int64_t c = part * 256 + threadIdx.x; c < bytes / 16; c += 256 * parts)
                *(v4i *)(d + c * 16) = *(const v4i *)(s + c * 16);
        } else {
            for (int64_t c = part * 256 + threadIdx.x; c < bytes; c += 256 * parts) d[c] = s[c];
        }
    };
    run(d0 + dr * d0_stride, s0 + sr * s0_stride, bytes0);
    run(d1 + dr * d1_stride, s1 + sr * s1_stride, bytes1);
}

// ---- token routing for the dispatch (one pass of atomics instead of a device sort + histogram) ------------------
// rank[t] = how many tokens of the same expert were counted before token t (arrival order of the atomics: the order
// inside an expert's segment is unspecified, which the exchange does not care about -- dispatch and combine use the
// same table); counts[g] = tokens of expert g.
__global__ void __launch_bounds__(256) route_rank_kernel(const int64_t *ids, int64_t tokens, int groups,
                                                         unsigned long long *counts, int64_t *pos)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tokens) return;
    const int64_t e = ids[t];
    pos[t] = (e >= 0 && e < groups) ? (int64_t)atomicAdd(counts + e, 1ull) : -1;
}

// pos[t] = (exclusive prefix sum of counts)[ids[t]] + rank[t].  Every block rebuilds the prefix sums in LDS (groups is
// a few hundred at most), so the two steps need no third launch.
__global__ void __launch_bounds__(256) route_pos_kernel(const int64_t *ids, int64_t tokens, int groups,
                                                        const unsigned long long *counts, int64_t *pos)
{
    extern __shared__ long long off[];  // [groups] exclusive offsets, then [256] per-thread partial sums
    long long *partial = off + groups;
    const int per = (groups + 255) / 256, g0 = threadIdx.x * per, g1 = min(groups, g0 + per);
    long long sum = 0;
    for (int g = g0; g < g1; ++g) sum += (long long)counts[g];
    partial[threadIdx.x] = sum;
    __syncthreads();
    long long base = 0;
    for (int i = 0; i < (int)threadIdx.x; ++i) base += partial[i];
    for (int g = g0; g < g1; ++g) { off[g] = base; base += (long long)counts[g]; }
    __syncthreads();
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= tokens) return;
    const int64_t e = ids[t];
    if (e >= 0 && e < groups) pos[t] += off[e];
}

// ---- capacity-bounded slot assignment: the host-sync-free routing of the sharded forward --------------------------
// Row r carries an int32 key (an expert id).  bucket(key) names a destination with room for `cap` rows; the row gets
// the next free slot of its bucket by one atomic, dest[r] = bucket * cap + slot.  A key outside [0, key_div * key_mul)
// (unused payload rows carry -1) gets dest -1 silently; a full bucket gets dest -1 and adds one to *overflow per dropped row.
//   key_sub == 0:  bucket = key / key_div                       tag = key % key_div
//   key_sub  > 0:  bucket = ((key % key_div) / key_sub) * key_mul + key / key_div     (chunk-major, then rank)
// Source side of the dispatch: key_div = experts per rank, bucket = (chunk, destination rank), cap = rows per pair and
// chunk, tag (the expert's index on its owner) is written into the payload row's header.  Receiving side: key_div = 1,
// bucket = local expert, cap = m_max, dest = slot in the masked [G_local, m_max] layout, counts = masked_m;
// inverse[dest] = r + inverse_base is the slot -> row table the indexed GEMM reads the rows through.
// One workgroup takes ROUTE_ROWS_PER_BLOCK rows: a private histogram in LDS gives every row its rank inside the
// block (LDS atomics), then ONE global atomic per (block, non-empty bucket) reserves the block's range -- with 128 rows
// per expert the per-row global atomics of a naive version serialise on 256 addresses (37 us for 32 768 tokens).
constexpr int ROUTE_THREADS = 1024, ROUTE_ROWS_PER_THREAD = 4, ROUTE_ROWS_PER_BLOCK = ROUTE_THREADS * ROUTE_ROWS_PER_THREAD;
__global__ void __launch_bounds__(ROUTE_THREADS) route_slots_kernel(const uint8_t *keys, int64_t key_stride, int64_t rows, int key_div,
                                                                    int key_sub, int key_mul, int buckets, int cap, int32_t *counts,
                                                                    int64_t *dest, uint8_t *tags, int64_t tag_stride, int32_t *overflow,
                                                                    int64_t *inverse, int64_t inverse_base)
{
    extern __shared__ int32_t route_lds[];   // [buckets] rows of this block per bucket, then [buckets] the block's first slot
    int32_t *cnt = route_lds, *base = route_lds + buckets;
    for (int b = threadIdx.x; b < buckets; b += ROUTE_THREADS) cnt[b] = 0;
    __syncthreads();
    int bucket[ROUTE_ROWS_PER_THREAD], lo[ROUTE_ROWS_PER_THREAD], rank[ROUTE_ROWS_PER_THREAD];
    const int64_t r0 = (int64_t)blockIdx.x * ROUTE_ROWS_PER_BLOCK + threadIdx.x;
#pragma unroll
    for (int i = 0; i < ROUTE_ROWS_PER_THREAD; ++i) {
        const int64_t r = r0 + (int64_t)i * ROUTE_THREADS;
        bucket[i] = -1;
        if (r < rows) {
            const int key = *(const int32_t *)(keys + r * key_stride);
            if (key >= 0) {
                const int hi = key / key_div;
                lo[i] = key - hi * key_div;
                const int bk = key_sub ? (lo[i] / key_sub) * key_mul + hi : hi;
                if (hi < (key_sub ? key_mul : buckets) && bk < buckets) {
                    bucket[i] = bk;
                    rank[i] = atomicAdd(cnt + bk, 1);
                }
            }
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < buckets; b += ROUTE_THREADS) {
        const int c = cnt[b];
        if (c > 0) {
            const int old = atomicAdd(counts + b, c);
            base[b] = old;
            if (old + c > cap) {   // the tail of this block's rows does not fit: give the slots back, count the dropped rows
                const int dropped = min(c, old + c - cap);
                atomicSub(counts + b, dropped);
                atomicAdd(overflow, dropped);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ROUTE_ROWS_PER_THREAD; ++i) {
        const int64_t r = r0 + (int64_t)i * ROUTE_THREADS;
        if (r >= rows) continue;
        int64_t d = -1;
        if (bucket[i] >= 0) {
            const int slot = base[bucket[i]] + rank[i];
            if (slot < cap) {
                d = (int64_t)bucket[i] * cap + slot;
                if (tags) *(int32_t *)(tags + d * tag_stride) = lo[i];
                if (inverse) inverse[d] = r + inverse_base;
            }
        }
        dest[r] = d;
    }
}

}  // namespace dga

extern "C" int dga_route_slots(const void *keys, int64_t key_stride_bytes, int64_t rows, int key_div, int key_sub, int key_mul,
                               int buckets, int cap, int32_t *counts, int zero_counts, int64_t *dest, void *tags,
                               int64_t tag_stride_bytes, int32_t *overflow, int64_t *inverse, int64_t inverse_base,
                               void *stream)
{
    if (rows < 0 || key_div < 1 || key_sub < 0 || buckets < 0 || cap < 0 || key_stride_bytes < 4) return DGA_E_SHAPE;
    if (key_sub > 0 && key_mul < 1) return DGA_E_SHAPE;
    if (buckets > 0 && !counts) return DGA_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (zero_counts && buckets > 0 &&
        dga::record_hip(hipMemsetAsync(counts, 0, sizeof(int32_t) * buckets, st)) != DGA_OK)
        return DGA_E_HIP;
    if (rows == 0) return DGA_OK;
    if (!keys || !dest || !overflow) return DGA_E_NULL;
    if ((reinterpret_cast<uintptr_t>(keys) | static_cast<uintptr_t>(key_stride_bytes)) & 3) return DGA_E_ALIGN;
    if (tags && ((reinterpret_cast<uintptr_t>(tags) | static_cast<uintptr_t>(tag_stride_bytes)) & 3)) return DGA_E_ALIGN;
    if (buckets > 4096) return DGA_E_RANGE;   // two int32 per bucket in LDS
    hipLaunchKernelGGL(dga::route_slots_kernel,
                       dim3(static_cast<unsigned>((rows + dga::ROUTE_ROWS_PER_BLOCK - 1) / dga::ROUTE_ROWS_PER_BLOCK)),
                       dim3(dga::ROUTE_THREADS), sizeof(int32_t) * 2 * buckets, st,
                       static_cast<const uint8_t *>(keys), key_stride_bytes, rows, key_div, key_sub, key_mul, buckets, cap,
                       counts, dest, static_cast<uint8_t *>(tags), tag_stride_bytes, overflow, inverse, inverse_base);
    return dga::record_hip(hipGetLastError());
}

extern "C" int dga_route_tokens(const int64_t *expert_ids, int64_t tokens, int groups, int64_t *counts, int64_t *pos,
                                void *stream)
{
    if (tokens < 0 || groups < 0 || groups > 4096) return DGA_E_SHAPE;
    if (groups > 0 && !counts) return DGA_E_NULL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (groups > 0 && dga::record_hip(hipMemsetAsync(counts, 0, sizeof(int64_t) * groups, st)) != DGA_OK) return DGA_E_HIP;
    if (tokens == 0) return DGA_OK;
    if (!expert_ids || !pos) return DGA_E_NULL;
    const unsigned grid = static_cast<unsigned>((tokens + 255) / 256);
    hipLaunchKernelGGL(dga::route_rank_kernel, dim3(grid), dim3(256), 0, st, expert_ids, tokens, groups,
                       reinterpret_cast<unsigned long long *>(counts), pos);
    hipLaunchKernelGGL(dga::route_pos_kernel, dim3(grid), dim3(256), sizeof(long long) * (groups + 256), st, expert_ids,
                       tokens, groups, reinterpret_cast<const unsigned long long *>(counts), pos);
    return dga::record_hip(hipGetLastError());
}

namespace dga {

// Rows of src_row_bytes bytes at arbitrary byte alignment -> rows of dst_row_bytes (a multiple of 16, >= src_row_bytes, 16-byte
// aligned destination) with a zero-filled tail: the re-layout pass in front of the tile kernels when K does not cut into 16-byte DMA
// chunks (the CDNA4 reading of the reference's PaddingCommon variant, op_kernel/kernel/padding_common_matmul_kernel.h:33-107).
// HBM-bound byte work: one thread per 16-byte destination chunk; the source is read as ALIGNED dwords (five of them cover any 16
// bytes) funnel-shifted into place by v_alignbyte_b32, so a wave reads 1 KB of consecutive bytes per pass whatever the row's
// alignment.  Two operands (A and B of one GEMM) share a launch.  Only dwords that overlap the operand are touched (an aligned dword
// that holds one byte of an allocation lies inside it).
struct PadOperand {
    const uint8_t *src;
    uint8_t *dst;
    int64_t rows;
    int64_t stride;   // bytes between source rows (>= the row's own bytes)
};
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v4u_a4 __attribute__((ext_vector_type(4), aligned(4)));

__global__ void __launch_bounds__(256) pad_rows_kernel(PadOperand o0, PadOperand o1, int64_t chunks0, int64_t chunks, int src_row_bytes,
                                                       int dst_row_bytes)
{
    int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (chunk >= chunks) return;
    const bool second = chunk >= chunks0;
    const PadOperand o = second ? o1 : o0;
    if (second) chunk -= chunks0;
    const int cpr = dst_row_bytes >> 4;
    const int64_t r = chunk / cpr;
    const int c0 = (int)(chunk - r * cpr) << 4;
    const int valid = src_row_bytes - c0;   // bytes of this row at and after c0
    v4u w = {0u, 0u, 0u, 0u};
    if (valid > 0) {
        const uint8_t *s = o.src + r * o.stride + c0;
        const uint32_t sh = (uint32_t)(reinterpret_cast<uintptr_t>(s) & 3);
        const uint32_t *al = reinterpret_cast<const uint32_t *>(s - sh);
        const uint8_t *end = o.src + (o.rows - 1) * o.stride + src_row_bytes;
        uint32_t d[5];
        if (reinterpret_cast<const uint8_t *>(al + 5) <= end) {
            const v4u q = *reinterpret_cast<const v4u_a4 *>(al);
            d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; d[4] = al[4];
        } else {
#pragma unroll
            for (int j = 0; j < 5; ++j) d[j] = (reinterpret_cast<const uint8_t *>(al + j) < end) ? al[j] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t x = __builtin_amdgcn_alignbyte(d[j + 1], d[j], sh);
            const int nb = valid - 4 * j;   // bytes of word j that belong to the row
            if (nb < 4) x = nb <= 0 ? 0u : (x & ((1u << (8 * nb)) - 1u));
            w[j] = x;
        }
    }
    *reinterpret_cast<v4u *>(o.dst + r * dst_row_bytes + c0) = w;
}

int pad_rows(const void *src0, void *dst0, int64_t rows0, const void *src1, void *dst1, int64_t rows1, int64_t src_row_bytes,
             int64_t dst_row_bytes, hipStream_t stream)
{
    return pad_rows_strided(src0, src_row_bytes, dst0, rows0, src1, src_row_bytes, dst1, rows1, src_row_bytes, dst_row_bytes, stream);
}

int pad_rows_strided(const void *src0, int64_t stride0, void *dst0, int64_t rows0, const void *src1, int64_t stride1, void *dst1,
                     int64_t rows1, int64_t src_row_bytes, int64_t dst_row_bytes, hipStream_t stream)
{
    if (rows0 < 0 || rows1 < 0 || src_row_bytes < 0 || dst_row_bytes < src_row_bytes || (dst_row_bytes & 15) ||
        dst_row_bytes > 0x7FFFFFF0ll || (rows0 && stride0 < src_row_bytes) || (rows1 && stride1 < src_row_bytes))
        return DGA_E_SHAPE;
    if (((reinterpret_cast<uintptr_t>(dst0) | reinterpret_cast<uintptr_t>(dst1)) & 15) != 0) return DGA_E_ALIGN;
    const int64_t cpr = dst_row_bytes >> 4;
    const int64_t chunks0 = rows0 * cpr, chunks = chunks0 + rows1 * cpr;
    if (chunks == 0) return DGA_OK;
    if ((rows0 && (!src0 || !dst0)) || (rows1 && (!src1 || !dst1))) return DGA_E_NULL;
    if ((chunks + 255) / 256 > 0x7FFFFFFFll) return DGA_E_RANGE;
    hipLaunchKernelGGL(pad_rows_kernel, dim3(static_cast<unsigned>((chunks + 255) / 256)), dim3(256), 0, stream,
                       PadOperand{static_cast<const uint8_t *>(src0), static_cast<uint8_t *>(dst0), rows0, stride0},
                       PadOperand{static_cast<const uint8_t *>(src1), static_cast<uint8_t *>(dst1), rows1, stride1}, chunks0, chunks,
                       static_cast<int>(src_row_bytes), static_cast<int>(dst_row_bytes));
    return record_hip(hipGetLastError());
}

}  // namespace dga

extern "C" int dga_copy_rows2(void *dst0, int64_t dst0_row_stride, const void *src0, int64_t src0_row_stride, int64_t row_bytes0,
                              void *dst1, int64_t dst1_row_stride, const void *src1, int64_t src1_row_stride, int64_t row_bytes1,
                              const int64_t *dst_index, const int64_t *src_index, int64_t rows, void *stream)
{
    if (rows < 0 || row_bytes0 < 0 || row_bytes1 < 0) return DGA_E_SHAPE;
    if (rows == 0 || (row_bytes0 == 0 && row_bytes1 == 0)) return DGA_OK;
    if ((row_bytes0 && (!dst0 || !src0)) || (row_bytes1 && (!dst1 || !src1))) return DGA_E_NULL;
    const int vec = ((reinterpret_cast<uintptr_t>(dst0) | reinterpret_cast<uintptr_t>(src0) | reinterpret_cast<uintptr_t>(dst1) |
                      reinterpret_cast<uintptr_t>(src1) | dst0_row_stride | src0_row_stride | dst1_row_stride |
                      src1_row_stride | row_bytes0 | row_bytes1) & 15) == 0;
    const int64_t big = row_bytes0 > row_bytes1 ? row_bytes0 : row_bytes1;
    int parts = static_cast<int>((big / (vec ? 16 : 1) + 256 * 4 - 1) / (256 * 4));
    if (parts < 1) parts = 1;
    if (parts > 64) parts = 64;
    if (rows * parts > 0x7FFFFFFFll) return DGA_E_RANGE;
    hipLaunchKernelGGL(dga::copy_rows2_kernel, dim3(static_cast<unsigned>(rows * parts)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<uint8_t *>(dst0), dst0_row_stride,
                       static_cast<const uint8_t *>(src0), src0_row_stride, row_bytes0, static_cast<uint8_t *>(dst1),
                       dst1_row_stride, static_cast<const uint8_t *>(src1), src1_row_stride, row_bytes1, dst_index, src_index,
                       rows, parts, vec);
    return dga::record_hip(hipGetLastError());
}

extern "C" int dga_copy_rows(void *dst, int64_t dst_row_stride, const int64_t *dst_index, const void *src,
                             int64_t src_row_stride, const int64_t *src_index, int64_t row_bytes, int64_t rows,
                             void *stream)
{
    if (rows < 0 || row_bytes < 0) return DGA_E_SHAPE;
    if (rows == 0 || row_bytes == 0) return DGA_OK;
    if (!dst || !src) return DGA_E_NULL;
    const bool vec = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | dst_row_stride |
                       src_row_stride | row_bytes) & 15) == 0;
    int parts = static_cast<int>((row_bytes / (vec ? 16 : 1) + 256 * 4 - 1) / (256 * 4));  // <= 4 chunks per thread
    if (parts < 1) parts = 1;
    if (parts > 64) parts = 64;
    if (rows * parts > 0x7FFFFFFFll) return DGA_E_RANGE;
    dim3 grid(static_cast<unsigned>(rows * parts));
    if (vec)
        hipLaunchKernelGGL(dga::copy_rows_kernel<true>, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<uint8_t *>(dst), dst_row_stride, dst_index, static_cast<const uint8_t *>(src),
                           src_row_stride, src_index, row_bytes, rows, parts);
    else
        hipLaunchKernelGGL(dga::copy_rows_kernel<false>, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<uint8_t *>(dst), dst_row_stride, dst_index, static_cast<const uint8_t *>(src),
                           src_row_stride, src_index, row_bytes, rows, parts);
    return dga::record_hip(hipGetLastError());
}
