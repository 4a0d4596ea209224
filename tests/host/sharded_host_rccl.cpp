// The C++ host of the expert-sharded forward over a REAL collective: INTEGRATION.md section 6 with its RCCL callback compiled and run,
// one process per GPU (test infrastructure, not product).  north_star: "partition the grouped / masked-M MoE GEMM across the GPUs of
// one node with an RCCL all-to-all over xGMI for expert sharding", "host code in C++".  The reference has no collective at all
// (/root/reference/deep_gemm_ascend/benchmark_msprof/main.cpp:24-26: one device, one stream).
//
// Every rank: ncclCommInitRank (the unique id travels through a file rank 0 writes) -> dga_sharded_layout -> hipMalloc by the layout's
// sizes -> dga_sharded_events_create -> dga_sharded_forward, whose two exchanges call all_to_all() below ON THE STREAM THE LIBRARY
// NAMES: ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd -- equal splits, every shape static, no count exchange, no
// host synchronisation.  Strict policy, indexed rows and the packed layout, one and two chunks, twice in a row; every result row of
// this rank's tokens against the CPU oracle, byte for byte.
//
//   usage: sharded_host_rccl <world> <rank> <id file>      exit code 0 = every case passed, 77 = fewer than <world> GPUs visible
#include <rccl/rccl.h>

#include <chrono>
#include <fstream>

#include "sharded_host_common.hpp"

#define NCCL_OK(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { fprintf(stderr, "%s:%d rccl error %d (%s)\n", __FILE__, __LINE__, (int)r_, ncclGetErrorString(r_)); exit(4); } } while (0)

struct Comm { ncclComm_t comm; int world; };

// the collective of INTEGRATION.md section 6
static int all_to_all(void *user, int /*direction*/, int /*chunk*/, const void *send, void *recv, size_t bytes_per_peer, void *stream)
{
    Comm *c = static_cast<Comm *>(user);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (ncclGroupStart() != ncclSuccess) return 1;
    for (int p = 0; p < c->world; ++p) {
        if (ncclSend(static_cast<const char *>(send) + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, p, c->comm, s) != ncclSuccess) return 1;
        if (ncclRecv(static_cast<char *>(recv) + (size_t)p * bytes_per_peer, bytes_per_peer, ncclUint8, p, c->comm, s) != ncclSuccess) return 1;
    }
    return ncclGroupEnd() == ncclSuccess ? 0 : 1;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: sharded_host_rccl <world> <rank> <id file>\n"); return 64; }
    const int world = atoi(argv[1]), rank = atoi(argv[2]);
    const char *id_path = argv[3];
    if (world < 2 || world > G || G % world || rank < 0 || rank >= world) { fprintf(stderr, "world must divide %d, 0 <= rank < world\n", G); return 64; }
    if (dga_abi_version() != DGA_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
    int devices = 0;
    if (hipGetDeviceCount(&devices) != hipSuccess || devices < world) { printf("SKIP: %d GPU(s) visible, %d needed\n", devices, world); return 77; }
    HIP_OK(hipSetDevice(rank));
    // ---- the communicator: rank 0 draws the id and publishes it (write, then rename: a reader never sees half a file)
    ncclUniqueId id;
    if (rank == 0) {
        NCCL_OK(ncclGetUniqueId(&id));
        const std::string tmp = std::string(id_path) + ".tmp";
        { std::ofstream f(tmp, std::ios::binary); f.write(reinterpret_cast<const char *>(&id), sizeof id); }
        if (rename(tmp.c_str(), id_path) != 0) { perror("rename"); return 2; }
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            std::ifstream f(id_path, std::ios::binary);
            if (f && f.read(reinterpret_cast<char *>(&id), sizeof id)) break;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) { fprintf(stderr, "rank %d: no id file after 120 s\n", rank); return 2; }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
    }
    Comm c{nullptr, world};
    NCCL_OK(ncclCommInitRank(&c.comm, world, id, rank));

    int failures = 0, cases = 0;
    const Problem p = make_problem(world, false);
    for (int indexed = 1; indexed >= 0; --indexed)
        for (int chunks = 1; chunks <= 2; ++chunks) {
            std::vector<uint16_t> got;
            int dropped = -1;
            const int rc = run_rank(p, world, rank, /*device*/rank, indexed, chunks, all_to_all, &c, &got, &dropped);
            ++cases;
            const std::vector<uint16_t> want = expected(p, rank, nullptr);
            size_t bad = 0;
            for (size_t i = 0; i < want.size(); ++i) bad += got[i] != want[i];
            const bool ok = rc == 0 && bad == 0 && dropped == 0;
            printf("rccl world %d rank %d indexed %d chunks %d: %zu rows, %zu values differ from the oracle, %d dropped -> %s\n", world, rank,
                   indexed, chunks, p.ids[rank].size(), bad, dropped, ok ? "ok" : "FAIL");
            failures += !ok;
        }
    HIP_OK(hipDeviceSynchronize());
    NCCL_OK(ncclCommDestroy(c.comm));
    printf("rank %d: %d of %d cases passed\n", rank, cases - failures, cases);
    return failures ? 1 : 0;
}
