// What the two C++ hosts of the expert-sharded forward share (tests/host/sharded_host.cpp: world 2 emulated on one device;
// tests/host/sharded_host_rccl.cpp: one process per GPU over RCCL): the problem, one rank's set-up and forward as INTEGRATION.md
// section 6 lists them, and the expected rows through the CPU oracle (the checker).  Test infrastructure, not product.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "dga_hip.h"

extern "C" int dga_oracle_gemm_fp8_fp8_bf16_nt(const uint8_t *a, const float *sfa, const uint8_t *b, const float *sfb, uint16_t *out,
                                               int64_t m, int64_t n, int64_t k);

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d hip error %d\n", __FILE__, __LINE__, (int)e_); exit(2); } } while (0)
#define DGA_OK_(x) do { int s_ = (x); if (s_ != DGA_OK) { fprintf(stderr, "%s:%d dga status %d (%s)\n", __FILE__, __LINE__, s_, dga_status_string(s_)); exit(3); } } while (0)

static constexpr int G = 8, M_MAX = 64, N = 256, K = 512, KB = K / 128, NB = N / 128;

struct Problem {
    std::vector<uint8_t> b;      // [G][N][K]
    std::vector<float> sfb;      // [G][NB][KB]
    std::vector<std::vector<uint8_t>> q;    // per rank [T][K]
    std::vector<std::vector<float>> sf;     // per rank [T][KB]
    std::vector<std::vector<int64_t>> ids;  // per rank [T]
};

static Problem make_problem(int world, bool overflow)
{
    Problem p;
    std::mt19937 rng(3);
    p.b.resize((size_t)G * N * K); p.sfb.resize((size_t)G * NB * KB);
    for (auto &v : p.b) v = (uint8_t)(rng() % 120);
    for (auto &v : p.sfb) v = 0.5f + (rng() % 1000) / 1000.0f;
    p.q.resize(world); p.sf.resize(world); p.ids.resize(world);
    for (int r = 0; r < world; ++r) {
        const int T = overflow ? M_MAX + 40 : 90 + 13 * r;
        p.q[r].resize((size_t)T * K); p.sf[r].resize((size_t)T * KB); p.ids[r].resize(T);
        for (auto &v : p.q[r]) v = (uint8_t)(rng() % 120);
        for (auto &v : p.sf[r]) v = 0.5f + (rng() % 1000) / 1000.0f;
        for (int t = 0; t < T; ++t) {
            int64_t g = overflow ? (t % 7 == 0 ? 4 : 2) : (int64_t)(rng() % G);
            if (!overflow && g == 5) g = 6;      // one expert receives nothing
            p.ids[r][t] = g;
        }
    }
    return p;
}

template <class T> static T *dmalloc(size_t bytes) { void *p = nullptr; HIP_OK(hipMalloc(&p, bytes ? bytes : 16)); HIP_OK(hipMemset(p, 0, bytes ? bytes : 16)); return static_cast<T *>(p); }

// one rank: build everything INTEGRATION.md section 6 lists, run the forward twice, return the result rows and the drop count
typedef int (*collective_fn)(void *user, int direction, int chunk, const void *send, void *recv, size_t bytes_per_peer, void *stream);
static int run_rank(const Problem &p, int world, int rank, int device, int indexed, int chunks, collective_fn all_to_all, void *user,
                    std::vector<uint16_t> *out, int *dropped)
{
    HIP_OK(hipSetDevice(device));
    const int T = (int)p.ids[rank].size(), gl = G / world;
    dga_sharded_shape_t sh{world, rank, G, M_MAX, N, K, chunks, /*max_tokens*/128, /*capacity_factor*/0.f, indexed, DGA_POLICY_STRICT};
    dga_sharded_layout_t lay;
    DGA_OK_(dga_sharded_layout(&sh, &lay));
    if (lay.indexed != indexed) { fprintf(stderr, "layout changed the indexed flag\n"); return 1; }
    dga_sharded_buffers_t buf{};
    buf.send = dmalloc<void>(lay.send_bytes); buf.recv = dmalloc<void>(lay.recv_bytes);
    buf.osend = dmalloc<void>(lay.osend_bytes); buf.oback = dmalloc<void>(lay.oback_bytes);
    buf.slot = dmalloc<int64_t>(lay.slot_bytes); buf.rdest = dmalloc<int64_t>(lay.rdest_bytes);
    buf.row_of_slot = dmalloc<int64_t>(lay.row_of_slot_bytes);
    buf.pair_cnt = dmalloc<int32_t>(lay.pair_cnt_bytes); buf.masked_m = dmalloc<int32_t>(lay.masked_m_bytes);
    buf.overflow = dmalloc<int32_t>(4);
    buf.packed_a = dmalloc<void>(lay.packed_a_bytes); buf.packed_sfa = dmalloc<float>(lay.packed_sfa_bytes);
    buf.packed_out = dmalloc<void>(lay.packed_out_bytes);
    uint8_t *db = dmalloc<uint8_t>((size_t)gl * N * K); float *dsfb = dmalloc<float>((size_t)gl * NB * KB * 4);
    HIP_OK(hipMemcpy(db, p.b.data() + (size_t)rank * gl * N * K, (size_t)gl * N * K, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dsfb, p.sfb.data() + (size_t)rank * gl * NB * KB, (size_t)gl * NB * KB * 4, hipMemcpyHostToDevice));
    buf.b = db; buf.sfb = dsfb;
    uint8_t *dq = dmalloc<uint8_t>((size_t)T * K); float *dsf = dmalloc<float>((size_t)T * KB * 4); int64_t *dids = dmalloc<int64_t>((size_t)T * 8);
    uint16_t *dres = dmalloc<uint16_t>((size_t)T * N * 2);
    HIP_OK(hipMemcpy(dq, p.q[rank].data(), (size_t)T * K, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dsf, p.sf[rank].data(), (size_t)T * KB * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(dids, p.ids[rank].data(), (size_t)T * 8, hipMemcpyHostToDevice));
    hipStream_t st[3];
    for (auto &s : st) HIP_OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    void *streams[3] = {st[0], st[1], st[2]};
    std::vector<void *> events(lay.events > 0 ? lay.events : 1);
    if (world > 1) DGA_OK_(dga_sharded_events_create(lay.events, events.data()));
    for (int rep = 0; rep < 2; ++rep) {
        HIP_OK(hipMemsetAsync(dres, 0x7F, (size_t)T * N * 2, st[0]));     // dirty: dropped rows must be WRITTEN as zeros
        DGA_OK_(dga_sharded_forward(&sh, &buf, dq, dsf, dids, T, dres, 0, streams, world > 1 ? events.data() : nullptr,
                                    world > 1 ? all_to_all : nullptr, user));
    }
    for (auto &s : st) HIP_OK(hipStreamSynchronize(s));
    out->resize((size_t)T * N);
    HIP_OK(hipMemcpy(out->data(), dres, (size_t)T * N * 2, hipMemcpyDeviceToHost));
    int32_t ov = 0;
    HIP_OK(hipMemcpy(&ov, buf.overflow, 4, hipMemcpyDeviceToHost));
    *dropped = ov;
    if (world > 1) DGA_OK_(dga_sharded_events_destroy(lay.events, events.data()));
    for (auto &s : st) HIP_OK(hipStreamDestroy(s));
    for (void *q : {buf.send, buf.recv, buf.osend, buf.oback, (void *)buf.slot, (void *)buf.rdest, (void *)buf.row_of_slot, (void *)buf.pair_cnt,
                    (void *)buf.masked_m, (void *)buf.overflow, buf.packed_a, (void *)buf.packed_sfa, buf.packed_out, (void *)db, (void *)dsfb,
                    (void *)dq, (void *)dsf, (void *)dids, (void *)dres})
        HIP_OK(hipFree(q));
    return 0;
}

// expected rows of one rank's tokens: every expert's rows through the CPU oracle (rows in token order within an expert)
static std::vector<uint16_t> expected(const Problem &p, int rank, const std::vector<char> *dropped_mask)
{
    const int T = (int)p.ids[rank].size();
    std::vector<uint16_t> want((size_t)T * N, 0);
    for (int g = 0; g < G; ++g) {
        std::vector<int> rows;
        for (int t = 0; t < T; ++t)
            if (p.ids[rank][t] == g && !(dropped_mask && (*dropped_mask)[t])) rows.push_back(t);
        if (rows.empty()) continue;
        std::vector<uint8_t> a(rows.size() * K); std::vector<float> sfa(rows.size() * KB); std::vector<uint16_t> o(rows.size() * N);
        for (size_t i = 0; i < rows.size(); ++i) {
            memcpy(&a[i * K], &p.q[rank][(size_t)rows[i] * K], K);
            memcpy(&sfa[i * KB], &p.sf[rank][(size_t)rows[i] * KB], KB * 4);
        }
        dga_oracle_gemm_fp8_fp8_bf16_nt(a.data(), sfa.data(), &p.b[(size_t)g * N * K], &p.sfb[(size_t)g * NB * KB], o.data(), (int64_t)rows.size(), N, K);
        for (size_t i = 0; i < rows.size(); ++i) memcpy(&want[(size_t)rows[i] * N], &o[i * N], N * 2);
    }
    return want;
}

