// A C++ host of the expert-sharded forward: section 6 of INTEGRATION.md compiled and run (test infrastructure, not product).
//
// The reference's host language is C++ (framework/csrc/python_api.cpp, benchmark_msprof/main.cpp); this program is what such a
// host writes against include/dga_hip.h -- dga_sharded_layout -> hipMalloc by the layout's sizes -> dga_sharded_events_create ->
// dga_sharded_forward with its own collective -- with no Python, no torch and no ctypes in between.  It runs
//   world 1: no exchange (all_to_all = NULL);
//   world 2, emulated on ONE device: two host threads are the two ranks, each with its own buffers, streams and events; the
//            collective callback copies the peers' slices device-to-device, ORDERED BY EVENTS on the stream the library names
//            (no device synchronisation anywhere, as around a real RCCL call) -- the C++ twin of tests/test_parallel_gpu.py's
//            _FakeDist;
// under the strict policy, with indexed rows and with the packed layout, one and two chunks, twice in a row (static buffers
// reused), and compares every result row with the CPU oracle (oracle/libdga_oracle.so, linked as the checker), byte for byte.
// It also drives one overflowing forward and reads the dropped-row counter.
//
//   usage: sharded_host <world: 1|2>          exit code 0 = every case passed
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

struct Barrier {   // reusable host barrier of the rank threads
    std::mutex m; std::condition_variable cv; int n, waiting = 0, gen = 0;
    explicit Barrier(int n_) : n(n_) {}
    void wait() {
        std::unique_lock<std::mutex> l(m);
        const int g = gen;
        if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
        else cv.wait(l, [&] { return gen != g; });
    }
};

struct Exchange {   // what the ranks post for each other during one collective
    int world;
    Barrier bar;
    const void *send[2] = {nullptr, nullptr};
    hipEvent_t ready[2], done[2];
    explicit Exchange(int w) : world(w), bar(w) {
        for (int r = 0; r < 2; ++r) { HIP_OK(hipEventCreateWithFlags(&ready[r], hipEventDisableTiming)); HIP_OK(hipEventCreateWithFlags(&done[r], hipEventDisableTiming)); }
    }
};
struct RankCtx { Exchange *ex; int rank; };

// the collective of INTEGRATION.md section 6, with device copies in place of ncclSend / ncclRecv
static int all_to_all(void *user, int /*direction*/, int /*chunk*/, const void *send, void *recv, size_t bytes_per_peer, void *stream)
{
    RankCtx *c = static_cast<RankCtx *>(user);
    Exchange &ex = *c->ex;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (hipEventRecord(ex.ready[c->rank], s) != hipSuccess) return 1;      // my send slices are complete at this point of MY stream
    ex.send[c->rank] = send;
    ex.bar.wait();
    for (int src = 0; src < ex.world; ++src) {
        if (hipStreamWaitEvent(s, ex.ready[src], 0) != hipSuccess) return 1;   // the peer's slices are complete
        if (hipMemcpyAsync(static_cast<char *>(recv) + src * bytes_per_peer, static_cast<const char *>(ex.send[src]) + c->rank * bytes_per_peer,
                           bytes_per_peer, hipMemcpyDeviceToDevice, s) != hipSuccess) return 1;
    }
    if (hipEventRecord(ex.done[c->rank], s) != hipSuccess) return 1;       // I have read every peer's buffer
    ex.bar.wait();
    for (int src = 0; src < ex.world; ++src)                                // nobody overwrites a send buffer a peer still reads
        if (hipStreamWaitEvent(s, ex.done[src], 0) != hipSuccess) return 1;
    ex.bar.wait();
    return 0;
}

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
static int run_rank(const Problem &p, int world, int rank, int indexed, int chunks, Exchange *ex, std::vector<uint16_t> *out, int *dropped)
{
    HIP_OK(hipSetDevice(0));
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
    RankCtx ctx{ex, rank};
    for (int rep = 0; rep < 2; ++rep) {
        HIP_OK(hipMemsetAsync(dres, 0x7F, (size_t)T * N * 2, st[0]));     // dirty: dropped rows must be WRITTEN as zeros
        DGA_OK_(dga_sharded_forward(&sh, &buf, dq, dsf, dids, T, dres, 0, streams, world > 1 ? events.data() : nullptr,
                                    world > 1 ? all_to_all : nullptr, &ctx));
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

int main(int argc, char **argv)
{
    const int world = argc > 1 ? atoi(argv[1]) : 1;
    if (world != 1 && world != 2) { fprintf(stderr, "usage: sharded_host <1|2>\n"); return 64; }
    if (dga_abi_version() != DGA_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    int failures = 0, cases = 0;
    const Problem p = make_problem(world, false);
    for (int indexed = 1; indexed >= 0; --indexed)
        for (int chunks = 1; chunks <= 2; ++chunks) {
            Exchange ex(world);
            std::vector<std::vector<uint16_t>> got(world);
            std::vector<int> dropped(world, -1), rc(world, 0);
            std::vector<std::thread> th;
            for (int r = 0; r < world; ++r)
                th.emplace_back([&, r] { rc[r] = run_rank(p, world, r, indexed, chunks, &ex, &got[r], &dropped[r]); });
            for (auto &t : th) t.join();
            for (int r = 0; r < world; ++r) {
                ++cases;
                const std::vector<uint16_t> want = expected(p, r, nullptr);
                size_t bad = 0;
                for (size_t i = 0; i < want.size(); ++i) bad += got[r][i] != want[i];
                const bool ok = rc[r] == 0 && bad == 0 && dropped[r] == 0;
                printf("world %d rank %d indexed %d chunks %d: %zu rows, %zu values differ from the oracle, %d dropped -> %s\n", world, r, indexed,
                       chunks, p.ids[r].size(), bad, dropped[r], ok ? "ok" : "FAIL");
                failures += !ok;
            }
        }
    if (world == 1) {   // an expert that receives more rows than it holds: the surplus comes back as zero rows and is COUNTED
        const Problem po = make_problem(1, true);
        Exchange ex(1);
        std::vector<uint16_t> got; int dropped = -1;
        const int rc = run_rank(po, 1, 0, 1, 1, &ex, &got, &dropped);
        const int T = (int)po.ids[0].size();
        int to2 = 0;
        for (int t = 0; t < T; ++t) to2 += po.ids[0][t] == 2;
        // which rows were dropped is the atomics' arrival order: a dropped row is all zeros, the others match the oracle
        std::vector<char> mask(T, 0);
        int zero_rows = 0;
        for (int t = 0; t < T; ++t) {
            bool z = true;
            for (int j = 0; j < N; ++j) z = z && got[(size_t)t * N + j] == 0;
            if (z && po.ids[0][t] == 2) { mask[t] = 1; ++zero_rows; }
        }
        const std::vector<uint16_t> want = expected(po, 0, &mask);
        size_t bad = 0;
        for (size_t i = 0; i < want.size(); ++i) bad += got[i] != want[i];
        ++cases;
        // two forwards ran: the counter is sticky and holds both
        const bool ok = rc == 0 && bad == 0 && zero_rows == to2 - M_MAX && dropped == 2 * (to2 - M_MAX);
        printf("world 1 overflow: %d rows for one expert of %d, %d zero rows, counter %d after two forwards, %zu values differ -> %s\n", to2, M_MAX,
               zero_rows, dropped, bad, ok ? "ok" : "FAIL");
        failures += !ok;
    }
    printf("%d of %d cases passed\n", cases - failures, cases);
    return failures ? 1 : 0;
}
