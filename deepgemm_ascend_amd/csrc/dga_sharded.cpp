// The expert-sharded forward of the grouped masked-M GEMM behind the C ABI (BASELINE configs[4]).
//
// The reference has no collective and no routing of any kind ("multi-card" = independent processes,
// /root/reference/deep_gemm_ascend/benchmark_msprof/main.cpp:24-26, framework/benchmark/benchmark.py:249-253); its host
// language is C++ behind pybind11 (framework/csrc/python_api.cpp).  This file is what such a host calls:
//   dga_sharded_layout   sizes of the static exchange buffers, payload row format, pair capacity, plan length
//   dga_sharded_plan     the forward as a fixed sequence of steps: (operation, stream, event to wait for / to record, slice)
//   dga_sharded_forward  the executor: walks the plan, launches the device steps itself (dga_route_slots, dga_copy_rows[2],
//                        the grouped GEMM) on the caller's three streams, orders them with the caller's events, and hands
//                        the two collectives to a callback -- a host with an ncclComm_t runs ncclGroupStart / ncclSend /
//                        ncclRecv x peers / ncclGroupEnd (or ncclAllToAll) there; deepgemm_ascend_amd/parallel.py passes
//                        torch.distributed.all_to_all_single.
// Every shape is static and nothing is read back, so the whole forward captures into one HIP graph.  The plan is pure host
// arithmetic: parallel.py interprets the SAME plan with CPU tensors in the world-2 gloo tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "dga_hip.h"
#include "dga_internal.hpp"

namespace {

struct Derived {
    int gl, glc, chunks, kb, nb;
    int64_t hdr, row_bytes, cap, per, rows, max_tokens;
    bool indexed;
};

int derive(const dga_sharded_shape_t &s, Derived &d)
{
    if (s.world < 1 || s.rank < 0 || s.rank >= s.world || s.groups_total < 1 || s.m_max < 1 || s.n < 1 || s.k < 1) return DGA_E_SHAPE;
    if (s.groups_total % s.world) return DGA_E_SHAPE;             // experts must divide evenly over the ranks
    // a payload row is [K fp8 bytes][scales as floats][int32 header]: with more than one rank the routing steps read the
    // header (and the indexed GEMM the scales) in place, which needs K % 4 == 0 -- said here, not by a failing step halfway
    // through a forward (a single rank has no payload rows: any K)
    if (s.world > 1 && (s.k % 4) != 0) return DGA_E_SHAPE;
    d.gl = s.groups_total / s.world;
    int chunks = s.chunks;
    if (chunks <= 0) chunks = (s.world > 1 && d.gl % 2 == 0 && d.gl >= 8) ? 2 : 1;
    if (d.gl % chunks) return DGA_E_SHAPE;
    d.chunks = chunks;
    d.glc = d.gl / chunks;
    d.kb = (s.k + 127) / 128;
    d.nb = (s.n + 127) / 128;
    d.hdr = static_cast<int64_t>(s.k) + 4ll * d.kb;               // fp8 bytes, then the 1x128 scales, then the 4-byte header
    d.row_bytes = (d.hdr + 4 + 127) / 128 * 128;                  // rows start on 128-byte lines
    d.max_tokens = s.max_tokens > 0 ? s.max_tokens : static_cast<int64_t>(d.gl) * s.m_max;
    // rows reserved per (chunk, destination rank): min(tokens, experts of a chunk x m_max) cannot overflow before an expert
    // does; capacity_factor trades that guarantee for less padding on the wire
    const int64_t bound = std::max<int64_t>(1, std::min<int64_t>(d.max_tokens, static_cast<int64_t>(d.glc) * s.m_max));
    if (s.capacity_factor > 0.f) {
        const double even = static_cast<double>(d.max_tokens) / (static_cast<double>(s.world) * chunks);
        const int64_t want = (static_cast<int64_t>(std::ceil(static_cast<double>(s.capacity_factor) * even)) + 15) / 16 * 16;
        d.cap = std::min<int64_t>(bound, std::max<int64_t>(16, want));
    } else {
        d.cap = bound;
    }
    d.per = s.world > 1 ? static_cast<int64_t>(s.world) * d.cap : 0;
    d.rows = d.per * chunks;
    d.indexed = s.indexed != 0;
    // the tile loads address a row's scales as floats inside the payload row, and everything with 32-bit byte offsets
    if (d.indexed && s.world > 1 && d.rows * d.row_bytes >= 0x7FFFFFFFll) d.indexed = false;
    if (d.indexed && s.world == 1 && d.max_tokens * static_cast<int64_t>(s.k) >= 0x7FFFFFFFll) d.indexed = false;
    return DGA_OK;
}

void push(std::vector<dga_sharded_step_t> &v, int op, int stream, int chunk, int event, int64_t row_begin, int64_t rows,
          int group_begin, int groups)
{
    dga_sharded_step_t st{};
    st.op = op; st.stream = stream; st.chunk = chunk; st.event = event;
    st.row_begin = row_begin; st.rows = rows; st.group_begin = group_begin; st.groups = groups;
    v.push_back(st);
}

// events: 0 = "packed and counted" (main -> dispatch stream), 1 + c = chunk c received and routed (dispatch -> main),
// 1 + chunks + c = chunk c computed (main -> combine stream), 1 + 2 chunks + c = chunk c back (combine -> main)
int build_plan(const dga_sharded_shape_t &s, const Derived &d, std::vector<dga_sharded_step_t> &v)
{
    v.clear();
    if (s.world == 1) {   // no exchange: tokens are routed straight into (or addressed through) the masked layout
        push(v, DGA_STEP_ROUTE_SOURCE, 0, -1, -1, 0, 0, 0, d.gl);
        push(v, DGA_STEP_ZERO_DROPPED, 0, -1, -1, 0, 0, 0, 0);
        if (!d.indexed) push(v, DGA_STEP_PACK, 0, -1, -1, 0, 0, 0, 0);
        push(v, DGA_STEP_GEMM, 0, 0, -1, 0, 0, 0, d.gl);
        if (!d.indexed) push(v, DGA_STEP_RESTORE_ORDER, 0, -1, -1, 0, 0, 0, 0);
        return DGA_OK;
    }
    const int ch = d.chunks;
    push(v, DGA_STEP_CLEAR_HEADERS, 0, -1, -1, 0, d.rows, 0, 0);
    push(v, DGA_STEP_ROUTE_SOURCE, 0, -1, -1, 0, d.rows, 0, 0);
    push(v, DGA_STEP_PACK, 0, -1, -1, 0, d.rows, 0, 0);
    push(v, DGA_STEP_ZERO_COUNTS, 0, -1, -1, 0, 0, 0, d.gl);
    push(v, DGA_STEP_ZERO_DROPPED, 0, -1, -1, 0, 0, 0, 0);
    push(v, DGA_STEP_RECORD_EVENT, 0, -1, 0, 0, 0, 0, 0);
    push(v, DGA_STEP_WAIT_EVENT, 1, -1, 0, 0, 0, 0, 0);
    for (int c = 0; c < ch; ++c) {   // dispatch: exchange, then receive-side slots (+ the unpack copy of the packed path)
        push(v, DGA_STEP_ALL_TO_ALL_DISPATCH, 1, c, -1, c * d.per, d.per, 0, 0);
        push(v, DGA_STEP_ROUTE_RECEIVED, 1, c, -1, c * d.per, d.per, c * d.glc, d.glc);
        push(v, DGA_STEP_ZERO_UNROUTED, 1, c, -1, c * d.per, d.per, c * d.glc, d.glc);
        if (!d.indexed) push(v, DGA_STEP_UNPACK, 1, c, -1, c * d.per, d.per, c * d.glc, d.glc);
        push(v, DGA_STEP_RECORD_EVENT, 1, c, 1 + c, 0, 0, 0, 0);
    }
    for (int c = 0; c < ch; ++c) {   // the grouped GEMM of the chunk's experts
        push(v, DGA_STEP_WAIT_EVENT, 0, c, 1 + c, 0, 0, 0, 0);
        push(v, DGA_STEP_GEMM, 0, c, -1, c * d.per, d.per, c * d.glc, d.glc);
        push(v, DGA_STEP_RECORD_EVENT, 0, c, 1 + ch + c, 0, 0, 0, 0);
    }
    for (int c = 0; c < ch; ++c) {   // combine: rows back in arrival order, exchange
        push(v, DGA_STEP_WAIT_EVENT, 2, c, 1 + ch + c, 0, 0, 0, 0);
        if (!d.indexed) push(v, DGA_STEP_GATHER_OUT, 2, c, -1, c * d.per, d.per, c * d.glc, d.glc);
        push(v, DGA_STEP_ALL_TO_ALL_COMBINE, 2, c, -1, c * d.per, d.per, 0, 0);
        push(v, DGA_STEP_RECORD_EVENT, 2, c, 1 + 2 * ch + c, 0, 0, 0, 0);
    }
    for (int c = 0; c < ch; ++c) push(v, DGA_STEP_WAIT_EVENT, 0, c, 1 + 2 * ch + c, 0, 0, 0, 0);
    push(v, DGA_STEP_RESTORE_ORDER, 0, -1, -1, 0, d.rows, 0, 0);
    return DGA_OK;
}

// four bytes of `value` at base + r * stride (byte stores)
__global__ void fill_i32_strided_kernel(uint8_t *base, int64_t stride, int64_t rows, int32_t value)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    if (r < rows) {
        uint8_t *p = base + r * stride;
        for (int i = 0; i < 4; ++i) p[i] = static_cast<uint8_t>(static_cast<uint32_t>(value) >> (8 * i));
    }
}

// rows whose index entry is negative are zeroed (one wave per row; the others are left alone): the result row of a token that
// found no slot, the returning row of a received row that found no place in its expert -- instead of a memset of everything
__global__ void __launch_bounds__(256) zero_rows_where_negative_kernel(uint8_t *dst, int64_t row_stride, const int64_t *index,
                                                                       int64_t rows, int64_t row_bytes)
{
    const int64_t r = static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6);
    if (r >= rows || index[r] >= 0) return;
    uint8_t *p = dst + r * row_stride;
    const int lane = threadIdx.x & 63;
    if ((((uintptr_t)p | (uintptr_t)row_bytes) & 15) == 0) {
        for (int64_t o = lane * 16; o < row_bytes; o += 64 * 16) *reinterpret_cast<int4 *>(p + o) = int4{0, 0, 0, 0};
    } else {
        for (int64_t o = lane; o < row_bytes; o += 64) p[o] = 0;
    }
}

int zero_rows_where_negative(void *dst, int64_t row_stride, const int64_t *index, int64_t rows, int64_t row_bytes, hipStream_t s)
{
    if (rows <= 0) return DGA_OK;
    hipLaunchKernelGGL(zero_rows_where_negative_kernel, dim3(static_cast<unsigned>((rows + 3) / 4)), dim3(256), 0, s,
                       static_cast<uint8_t *>(dst), row_stride, index, rows, row_bytes);
    return dga::record_hip(hipGetLastError());
}

}  // namespace

extern "C" {

int dga_sharded_layout(const dga_sharded_shape_t *shape, dga_sharded_layout_t *out)
{
    if (!shape || !out) return DGA_E_NULL;
    Derived d{};
    if (int rc = derive(*shape, d)) return rc;
    std::memset(out, 0, sizeof(*out));
    out->groups_local = d.gl; out->groups_per_chunk = d.glc; out->chunks = d.chunks; out->kb = d.kb; out->nb = d.nb;
    out->indexed = d.indexed ? 1 : 0;
    out->hdr_offset = d.hdr; out->row_bytes = d.row_bytes; out->pair_capacity = shape->world > 1 ? d.cap : 0;
    out->rows_per_chunk = d.per; out->rows_total = d.rows; out->max_tokens = d.max_tokens;
    out->send_bytes = out->recv_bytes = static_cast<uint64_t>(d.rows) * d.row_bytes;
    out->osend_bytes = out->oback_bytes = static_cast<uint64_t>(d.rows) * shape->n * 2;
    out->slot_bytes = static_cast<uint64_t>(std::max<int64_t>(d.max_tokens, 1)) * 8;
    out->rdest_bytes = static_cast<uint64_t>(d.rows) * 8;
    out->row_of_slot_bytes = d.indexed ? static_cast<uint64_t>(d.gl) * shape->m_max * 8 : 0;
    out->pair_cnt_bytes = static_cast<uint64_t>(d.chunks) * shape->world * 4;
    out->masked_m_bytes = static_cast<uint64_t>(d.gl) * 4;
    if (!d.indexed) {
        out->packed_a_bytes = static_cast<uint64_t>(d.gl) * shape->m_max * shape->k;
        out->packed_sfa_bytes = static_cast<uint64_t>(d.gl) * shape->m_max * d.kb * 4;
        out->packed_out_bytes = static_cast<uint64_t>(d.gl) * shape->m_max * shape->n * 2;
    }
    out->events = shape->world > 1 ? 1 + 3 * d.chunks : 0;
    std::vector<dga_sharded_step_t> v;
    build_plan(*shape, d, v);
    out->steps = static_cast<int32_t>(v.size());
    return DGA_OK;
}

int dga_sharded_plan(const dga_sharded_shape_t *shape, dga_sharded_step_t *steps, int capacity, int *count)
{
    if (!shape || !count) return DGA_E_NULL;
    Derived d{};
    if (int rc = derive(*shape, d)) return rc;
    std::vector<dga_sharded_step_t> v;
    build_plan(*shape, d, v);
    *count = static_cast<int>(v.size());
    if (!steps) return DGA_OK;                         // size query
    if (capacity < *count) return DGA_E_WORKSPACE;
    std::memcpy(steps, v.data(), v.size() * sizeof(dga_sharded_step_t));
    return DGA_OK;
}

int dga_sharded_forward(const dga_sharded_shape_t *shape, const dga_sharded_buffers_t *buf, const void *tok_q,
                        const float *tok_sf, const int64_t *expert_ids, int tokens, void *result, int expected_m,
                        void *const *streams, void *const *events, dga_all_to_all_fn all_to_all, void *user)
{
    if (!shape || !buf || !streams) return DGA_E_NULL;
    Derived d{};
    if (int rc = derive(*shape, d)) return rc;
    if (tokens < 0 || tokens > d.max_tokens) return DGA_E_RANGE;
    if (tokens > 0 && (!tok_q || !tok_sf || !expert_ids || !result)) return DGA_E_NULL;
    if (!buf->b || !buf->sfb || !buf->slot || !buf->masked_m || !buf->overflow) return DGA_E_NULL;
    if (d.indexed && !buf->row_of_slot) return DGA_E_NULL;
    if (!d.indexed && (!buf->packed_a || !buf->packed_sfa || !buf->packed_out)) return DGA_E_NULL;
    if (shape->world > 1 && (!buf->send || !buf->recv || !buf->osend || !buf->oback || !buf->rdest || !buf->pair_cnt || !events ||
                             !all_to_all))
        return DGA_E_NULL;
    std::vector<dga_sharded_step_t> plan;
    build_plan(*shape, d, plan);
    const int n = shape->n, k = shape->k, m_max = shape->m_max, w = shape->world;
    const int em = expected_m > 0 ? expected_m : m_max;
    dga_tiling_t tiling{};
    {
        dga_problem_t pr{};
        pr.m = m_max; pr.n = n; pr.k = k; pr.groups = d.glc; pr.expected_m = em;
        pr.layoutTagA = DGA_LAYOUT_ROW_MAJOR; pr.layoutTagB = DGA_LAYOUT_COLUMN_MAJOR; pr.layoutTagC = DGA_LAYOUT_ROW_MAJOR;
        pr.dtype = DGA_DT_FP8_E4M3FN;
        // policy -1 = the library's default arithmetic ($DGA_DEFAULT_POLICY: bf16-exact unless told otherwise), -2 = the fast
        // policy's own tiling as it is; the bf16-exact policy picks its own tile (height from expected_m)
        // (-3 = the fast policy's tiling with DGA_POLICY_UE8M0_SCALES: the caller's promise of power-of-two scales)
        const int dp = dga::default_policy();
        if (shape->policy == -1 && dp < 0) return DGA_E_RANGE;     // $DGA_DEFAULT_POLICY names no policy
        int pol = shape->policy;
        if (pol == -1) {
            static const int of_default[] = {-2, DGA_POLICY_BF16_EXACT, DGA_POLICY_STRICT, -3, DGA_POLICY_BF16_EXACT | DGA_POLICY_UE8M0_SCALES, -2 /* auto: no decode kernel on this layout */};
            pol = of_default[dp];
        }
        if (pol < -3) return DGA_E_RANGE;
        const bool bx = pol >= 0 && (pol & 15) == DGA_POLICY_BF16_EXACT;
        if (int rc = bx ? dga_tiling_bf16_exact(&pr, &tiling) : dga_tiling(&pr, &tiling)) return rc;
        if (pol >= 0) tiling.dispatchPolicyTag = static_cast<uint8_t>(pol);
        else if (pol == -3) tiling.dispatchPolicyTag |= DGA_POLICY_UE8M0_SCALES;
    }
    uint8_t *send = static_cast<uint8_t *>(buf->send), *recv = static_cast<uint8_t *>(buf->recv);
    uint8_t *osend = static_cast<uint8_t *>(buf->osend), *oback = static_cast<uint8_t *>(buf->oback);
    const uint8_t *q = static_cast<const uint8_t *>(tok_q);
    const uint8_t *bw = static_cast<const uint8_t *>(buf->b);
    const int64_t b_gs = static_cast<int64_t>(n) * k, sfb_gs = static_cast<int64_t>(d.nb) * d.kb;
    for (const dga_sharded_step_t &st : plan) {
        void *s = streams[st.stream];
        hipStream_t hs = static_cast<hipStream_t>(s);
        int rc = DGA_OK;
        switch (st.op) {
        case DGA_STEP_WAIT_EVENT:
            rc = dga::record_hip(hipStreamWaitEvent(hs, static_cast<hipEvent_t>(events[st.event]), 0));
            break;
        case DGA_STEP_RECORD_EVENT:
            rc = dga::record_hip(hipEventRecord(static_cast<hipEvent_t>(events[st.event]), hs));
            break;
        case DGA_STEP_CLEAR_HEADERS:   // every header -1: rows nobody fills are skipped by the receiver
            if (st.rows > 0) {
                hipLaunchKernelGGL(fill_i32_strided_kernel, dim3(static_cast<unsigned>((st.rows + 255) / 256)), dim3(256), 0, hs,
                                   send + d.hdr, d.row_bytes, st.rows, -1);
                rc = dga::record_hip(hipGetLastError());
            }
            break;
        case DGA_STEP_ROUTE_SOURCE:
            if (w == 1)      // slot of every token in its expert's rows (+ the slot -> token table of the indexed GEMM)
                rc = dga_route_slots(expert_ids, 8, tokens, 1, 0, 1, d.gl, m_max, buf->masked_m, 1, buf->slot, nullptr, 0,
                                     buf->overflow, d.indexed ? buf->row_of_slot : nullptr, 0, s);
            else             // slot in the (chunk, destination) slice; header = the expert's index on its owner
                rc = dga_route_slots(expert_ids, 8, tokens, d.gl, d.glc, w, d.chunks * w, static_cast<int>(d.cap), buf->pair_cnt, 1,
                                     buf->slot, send + d.hdr, d.row_bytes, buf->overflow, nullptr, 0, s);
            break;
        case DGA_STEP_PACK:
            if (w == 1)
                rc = dga_copy_rows2(buf->packed_a, k, q, k, k, buf->packed_sfa, 4ll * d.kb, tok_sf, 4ll * d.kb, 4ll * d.kb, buf->slot,
                                    nullptr, tokens, s);
            else
                rc = dga_copy_rows2(send, d.row_bytes, q, k, k, send + k, d.row_bytes, tok_sf, 4ll * d.kb, 4ll * d.kb, buf->slot,
                                    nullptr, tokens, s);
            break;
        case DGA_STEP_ZERO_COUNTS:
            rc = dga::record_hip(hipMemsetAsync(buf->masked_m, 0, static_cast<size_t>(d.gl) * 4, hs));
            break;
        case DGA_STEP_ZERO_DROPPED:    // the row of a token that found no slot (a full bucket: *overflow says so) is never written
            rc = zero_rows_where_negative(result, 2ll * n, buf->slot, tokens, 2ll * n, hs);
            break;
        case DGA_STEP_ZERO_UNROUTED:   // ... and a received row that found no place in its expert travels back as zeros
            rc = zero_rows_where_negative(osend + st.row_begin * n * 2, 2ll * n, buf->rdest + st.row_begin, st.rows, 2ll * n, hs);
            break;
        case DGA_STEP_ALL_TO_ALL_DISPATCH:
            rc = all_to_all(user, 0, st.chunk, send + st.row_begin * d.row_bytes, recv + st.row_begin * d.row_bytes,
                            static_cast<size_t>(d.cap) * d.row_bytes, s);
            if (rc != 0) rc = DGA_E_HIP;
            break;
        case DGA_STEP_ROUTE_RECEIVED:  // per-expert row counts (= masked_m) by atomics, keyed by the received headers
            rc = dga_route_slots(recv + st.row_begin * d.row_bytes + d.hdr, d.row_bytes, st.rows, 1, 0, 1, d.gl, m_max, buf->masked_m, 0,
                                 buf->rdest + st.row_begin, nullptr, 0, buf->overflow, d.indexed ? buf->row_of_slot : nullptr,
                                 st.row_begin, s);
            break;
        case DGA_STEP_UNPACK:
            rc = dga_copy_rows2(buf->packed_a, k, recv + st.row_begin * d.row_bytes, d.row_bytes, k, buf->packed_sfa, 4ll * d.kb,
                                recv + st.row_begin * d.row_bytes + k, d.row_bytes, 4ll * d.kb, buf->rdest + st.row_begin, nullptr,
                                st.rows, s);
            break;
        case DGA_STEP_GEMM: {
            const int g0 = st.group_begin, g = st.groups;
            tiling.groups = static_cast<uint32_t>(g);
            if (d.indexed) {
                const void *a_src = w == 1 ? tok_q : static_cast<const void *>(recv);
                const float *sf_src = w == 1 ? tok_sf : reinterpret_cast<const float *>(recv + k);
                const int64_t lda = w == 1 ? k : d.row_bytes, sfa_ld = w == 1 ? d.kb : d.row_bytes / 4;
                const int64_t src_rows = w == 1 ? tokens : d.rows;
                void *dst = w == 1 ? result : static_cast<void *>(osend);
                rc = dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(
                    a_src, lda, sf_src, sfa_ld, bw + g0 * b_gs, buf->sfb + g0 * sfb_gs, dst, n,
                    buf->row_of_slot + static_cast<int64_t>(g0) * m_max, src_rows, buf->masked_m + g0, g, m_max, n, k, em, &tiling,
                    nullptr, 0, s);
            } else {
                rc = dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked(
                    static_cast<uint8_t *>(buf->packed_a) + static_cast<int64_t>(g0) * m_max * k,
                    buf->packed_sfa + static_cast<int64_t>(g0) * m_max * d.kb, bw + g0 * b_gs, buf->sfb + g0 * sfb_gs,
                    static_cast<uint8_t *>(buf->packed_out) + static_cast<int64_t>(g0) * m_max * n * 2, buf->masked_m + g0, g, m_max, n,
                    k, em, &tiling, buf->workspace, buf->workspace_bytes, s);
            }
            break;
        }
        case DGA_STEP_GATHER_OUT:      // result rows into the buffer that travels back, in arrival order
            rc = dga_copy_rows(osend + st.row_begin * n * 2, 2ll * n, nullptr, buf->packed_out, 2ll * n, buf->rdest + st.row_begin,
                               2ll * n, st.rows, s);
            break;
        case DGA_STEP_ALL_TO_ALL_COMBINE:
            rc = all_to_all(user, 1, st.chunk, osend + st.row_begin * n * 2, oback + st.row_begin * n * 2,
                            static_cast<size_t>(d.cap) * n * 2, s);
            if (rc != 0) rc = DGA_E_HIP;
            break;
        case DGA_STEP_RESTORE_ORDER:   // token order: result[t] = (returned | computed) row slot[t]
            rc = dga_copy_rows(result, 2ll * n, nullptr, w == 1 ? buf->packed_out : static_cast<const void *>(oback), 2ll * n, buf->slot,
                               2ll * n, tokens, s);
            break;
        default:
            rc = DGA_E_RANGE;
        }
        if (rc != DGA_OK) return rc;
    }
    return DGA_OK;
}

int dga_sharded_events_create(int count, void **events)
{
    if (count < 0 || (count > 0 && !events)) return DGA_E_NULL;
    for (int i = 0; i < count; ++i) {
        hipEvent_t e;
        if (int rc = dga::record_hip(hipEventCreateWithFlags(&e, hipEventDisableTiming))) {
            for (int j = 0; j < i; ++j) (void)hipEventDestroy(static_cast<hipEvent_t>(events[j]));
            return rc;
        }
        events[i] = e;
    }
    return DGA_OK;
}

int dga_sharded_events_destroy(int count, void **events)
{
    if (count < 0 || (count > 0 && !events)) return DGA_E_NULL;
    int rc = DGA_OK;
    for (int i = 0; i < count; ++i)
        if (events[i]) {
            const int r = dga::record_hip(hipEventDestroy(static_cast<hipEvent_t>(events[i])));
            if (r != DGA_OK) rc = r;
            events[i] = nullptr;
        }
    return rc;
}

}  // extern "C"
