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
#include "sharded_host_common.hpp"

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
                th.emplace_back([&, r] {
                    RankCtx ctx{&ex, r};
                    rc[r] = run_rank(p, world, r, /*device*/0, indexed, chunks, all_to_all, &ctx, &got[r], &dropped[r]);
                });
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
        RankCtx ctx{&ex, 0};
        const int rc = run_rank(po, 1, 0, /*device*/0, 1, 1, all_to_all, &ctx, &got, &dropped);
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
