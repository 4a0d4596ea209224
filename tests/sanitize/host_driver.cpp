// Host-side entry points of libdga_hip.so driven in bulk under AddressSanitizer + UBSan (tests/test_sanitizers.py builds the
// instrumented library with `make asan` and links this driver against it).  No GPU is needed: everything called here is host
// arithmetic -- operator hooks, tiling / kernel selection on both platform descriptions, the tiling cache with its CSV file
// (well-formed, reference-format and malformed), the learned predictor (default file, truncated file), the 28-int Config
// derivation, the sharded forward's layout and plan -- plus the argument checks of the launch entry points that return before
// any HIP call.  Exit code 0 and the final "ok" line = no sanitizer report (reports abort the process).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "dga_hip.h"

static int fails = 0;
#define CHECK(cond)                                                         \
    do {                                                                    \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); ++fails; } \
    } while (0)

static dga_problem_t problem(uint32_t m, uint32_t n, uint32_t k, uint32_t groups = 1, uint32_t expected_m = 0, uint32_t flags = 0)
{
    dga_problem_t p{};
    p.m = m; p.n = n; p.k = k; p.groups = groups; p.expected_m = expected_m; p.flags = flags;
    p.layoutTagA = DGA_LAYOUT_ROW_MAJOR; p.layoutTagB = DGA_LAYOUT_COLUMN_MAJOR; p.layoutTagC = DGA_LAYOUT_ROW_MAJOR;
    p.dtype = DGA_DT_FP8_E4M3FN;
    return p;
}

static void tiling_sweep(unsigned seed, int count)
{
    std::mt19937 rng(seed);
    const uint32_t edge[] = {0u, 1u, 2u, 7u, 15u, 16u, 17u, 127u, 128u, 129u, 255u, 256u, 1279u, 4096u, 7168u, 18432u, 65535u, 1u << 20, 0x7FFFFFFFu};
    dga_platform_t mi{}, a24{}, a20{};
    dga_platform_mi355x(&mi);
    dga_platform_ascend910b(&a24, 24);
    dga_platform_ascend910b(&a20, 20);
    for (int i = 0; i < count; ++i) {
        auto pick = [&] { return (rng() & 3) ? edge[rng() % (sizeof(edge) / sizeof(edge[0]))] : (rng() % 20000u); };
        const uint32_t m = pick(), n = pick(), k = pick();
        const uint32_t groups = (rng() & 7) == 0 ? 1 + rng() % 300 : 1;
        const uint32_t flags = (groups > 1 && (rng() & 1)) ? DGA_PROBLEM_CONTIGUOUS_M : 0;
        dga_problem_t p = problem(m, n, k, groups, (rng() & 1) ? rng() % 200 : 0, flags);
        dga_tiling_t t{};
        const int rc = dga_tiling(&p, &t);
        if (rc == DGA_OK) {
            CHECK(m == 0 || n == 0 || (t.m1 > 0 && t.n1 > 0));
            (void)dga_workspace_bytes(&t);
            float us = 0.f;
            (void)dga_predict_time_us(&p, &t, &us);
        }
        dga_tiling_t tb{};
        (void)dga_tiling_bf16_exact(&p, &tb);
        dga_tiling_t ts{};
        (void)dga_select_kernel(&p, &mi, &ts);
        p.dtype = DGA_DT_FP16;
        (void)dga_select_kernel(&p, &a24, &ts);
        (void)dga_select_kernel(&p, &a20, &ts);
        float pus = 0.f, nus = 0.f;
        p.dtype = DGA_DT_FP8_E4M3FN;
        (void)dga_select_kernel_with_predictor(&p, &ts, &pus, &nus);
        (void)dga_select_kernel_with_predictor_ex(&p, &ts, &pus, &nus, static_cast<int>(rng() % 3), static_cast<int>(rng() % 40));
    }
}

static std::string tmp(const char *name)
{
    const char *d = std::getenv("DGA_SAN_TMP");
    return std::string(d ? d : "/tmp") + "/" + name;
}

int main()
{
    CHECK(dga_abi_version() == DGA_ABI_VERSION);
    for (int s = -20; s <= 5; ++s) CHECK(dga_status_string(s) != nullptr);

    // ---- operator hooks
    {
        int64_t a[2] = {5, 7}, b[2] = {7, 9}, o[2] = {0, 0};
        CHECK(dga_infer_shape(a, 2, b, 2, o) == DGA_OK && o[0] == 5 && o[1] == 9);
        CHECK(dga_infer_shape(a, 1, b, 2, o) != DGA_OK);
        CHECK(dga_infer_shape(nullptr, 2, b, 2, o) != DGA_OK);
        int64_t bad[2] = {8, 9};
        CHECK(dga_infer_shape(a, 2, bad, 2, o) == DGA_OK);   // like the reference hook (catlass_dynamic_matmul.cpp:16-35): k is checked by the tiling hook
        int dt = -1;
        CHECK(dga_infer_dtype(DGA_DT_FP16, DGA_DT_FP16, &dt) == DGA_OK);
        CHECK(dga_infer_dtype(DGA_DT_FP16, DGA_DT_BF16, &dt) != DGA_OK);
        CHECK(dga_infer_dtype(DGA_DT_FP16, DGA_DT_FP16, nullptr) != DGA_OK);
    }

    // ---- tiling, memory-only cache
    CHECK(dga_tiling_cache_open(nullptr) == DGA_OK);
    CHECK(dga_tiling(nullptr, nullptr) != DGA_OK);
    tiling_sweep(1, 4000);
    CHECK(dga_tiling_cache_size() >= 0);
    CHECK(dga_tiling_cache_clear() == DGA_OK);

    // ---- CSV cache: append, reopen, reference-format file, malformed files
    {
        const std::string f = tmp("dga_san_cache.csv");
        std::remove(f.c_str());
        CHECK(dga_tiling_cache_open(f.c_str()) == DGA_OK);
        tiling_sweep(2, 600);
        const int n1 = dga_tiling_cache_size();
        CHECK(dga_tiling_cache_open(f.c_str()) == DGA_OK);   // re-read what was appended
        CHECK(dga_tiling_cache_size() > 0 && dga_tiling_cache_size() <= n1 + 1);
        tiling_sweep(2, 600);                                // all hits
        const std::string ref = tmp("dga_san_ref.csv");
        {
            std::ofstream o(ref);
            o << "m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim\n"
              << "4096,4096,4096,128,256,256,0,0,0,0,24\n128,2048,7168,128,96,512,0,0,0,0,22\n";
        }
        CHECK(dga_tiling_cache_open(ref.c_str()) == DGA_OK);
        tiling_sweep(3, 200);
        const std::string junk = tmp("dga_san_junk.csv");
        {
            std::ofstream o(junk);
            o << "m,n,k,m1,n1\n1,2\n,,,,,,,,,,,,,,,,,,,,,,,,,,\n4096,4096,4096,abc,def,ghi\n"
              << std::string(100000, 'x') << "\n-1,-2,-3,-4,-5,-6,-7,-8,-9,-10,-11\n99999999999999999999,1,1,1,1,1,1,1,1,1,1\n";
        }
        (void)dga_tiling_cache_open(junk.c_str());
        tiling_sweep(4, 200);
        (void)dga_tiling_cache_open(tmp("no/such/dir/cache.csv").c_str());
        CHECK(dga_tiling_cache_open(nullptr) == DGA_OK);
        std::remove(f.c_str()); std::remove(ref.c_str()); std::remove(junk.c_str());
    }

    // ---- the predictor's selection strategies on random lists (ties, one element, topk beyond the list, degenerate columns)
    {
        std::mt19937 rng(11);
        for (int it = 0; it < 4000; ++it) {
            const int n = 1 + static_cast<int>(rng() % 90);
            std::vector<float> preds(n);
            std::vector<int32_t> tiles(3 * n);
            for (int i = 0; i < n; ++i) {
                preds[i] = (it % 7 == 0) ? 5.0f : 1.0f + static_cast<float>(rng() % 1000) * ((it % 3) ? 0.37f : 0.0f);   // equal times too
                tiles[3 * i] = 16 << (rng() % 5); tiles[3 * i + 1] = (it % 5 == 0) ? 128 : 16 << (rng() % 5); tiles[3 * i + 2] = 64 << (rng() % 5);
            }
            for (int method = 0; method <= 2; ++method) {
                int picked = -1, cnt = -1;
                std::vector<int> members(n, -1);
                const int topk = static_cast<int>(rng() % 130) - 3;
                const int rc = dga_select_tiling_strategy(preds.data(), tiles.data(), n, method, topk, 0.1f + (rng() % 30) * 0.1f, 1 + static_cast<int>(rng() % 5),
                                                          rng() % 9, &picked, members.data(), &cnt);
                CHECK(rc == DGA_OK && picked >= 0 && picked < n && cnt >= 0 && cnt <= n);
                for (int i = 0; i < cnt; ++i) CHECK(members[i] >= 0 && members[i] < n);
            }
        }
        int picked = 0;
        float one = 1.f;
        CHECK(dga_select_tiling_strategy(nullptr, nullptr, 1, 0, 1, 0.8f, 2, 0, &picked, nullptr, nullptr) != DGA_OK);
        CHECK(dga_select_tiling_strategy(&one, nullptr, 1, DGA_PICK_TOPK_DBSCAN, 1, 0.8f, 2, 0, &picked, nullptr, nullptr) != DGA_OK);
        CHECK(dga_select_tiling_strategy(&one, nullptr, 0, 0, 1, 0.8f, 2, 0, &picked, nullptr, nullptr) != DGA_OK);
        CHECK(dga_select_tiling_strategy(&one, nullptr, 1, 9, 1, 0.8f, 2, 0, &picked, nullptr, nullptr) != DGA_OK);
        CHECK(dga_select_tiling_strategy(&one, nullptr, 1, DGA_PICK_GREEDY, 1, 0.8f, 2, 0, &picked, nullptr, nullptr) == DGA_OK && picked == 0);
    }

    // ---- predictor: default file, truncated copies, garbage
    {
        const int rc = dga_predictor_load(nullptr);
        CHECK(rc == DGA_OK || rc == DGA_E_IO);
        if (rc == DGA_OK) {
            CHECK(dga_predictor_loaded() == 1);
            tiling_sweep(5, 500);
        }
        const char *def = std::getenv("DGA_SAN_PREDICTOR");
        if (def) {
            std::ifstream in(def, std::ios::binary);
            std::string all((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
            for (size_t cut : {size_t(0), size_t(10), all.size() / 3, all.size() / 2, all.size() - 5}) {
                const std::string f = tmp("dga_san_pred.txt");
                { std::ofstream o(f, std::ios::binary); o.write(all.data(), static_cast<std::streamsize>(std::min(cut, all.size()))); }
                const int r = dga_predictor_load(f.c_str());
                CHECK(r == DGA_OK || r == DGA_E_IO);
                tiling_sweep(6, 50);
                std::remove(f.c_str());
            }
        }
        dga_predictor_unload();
        CHECK(dga_predictor_loaded() == 0);
        tiling_sweep(7, 100);
    }

    // ---- the 28-int Config derivation
    {
        std::mt19937 rng(11);
        uint32_t out[28];
        int32_t params[28];
        for (int i = 0; i < 3000; ++i) {
            const uint32_t m = rng() % 9000, n = rng() % 9000, k = rng() % 9000, b = rng() % 5;
            (void)dga_get_best_config(b, m, n, k, out);
            (void)dga_get_bench_config(m, n, k, rng() % 9, rng() % 9, rng() % 20, rng() % 20, rng() % 40, rng() % 20, out);
            std::memset(params, 0, sizeof(params));
            for (int j = 0; j < 6; ++j) params[j] = static_cast<int32_t>(rng() % 24);
            (void)dga_bench_params_fill(m, n, k, params);
            (void)dga_bbit_params(m, n, k, rng() % 9, rng() % 9, rng() % 20, rng() % 20, rng() % 40, rng() % 20, out);
        }
        CHECK(dga_get_best_config(1, 4096, 4096, 4096, nullptr) != DGA_OK);
    }

    // ---- sharded forward: layout and plan
    {
        std::mt19937 rng(13);
        for (int i = 0; i < 2000; ++i) {
            dga_sharded_shape_t s{};
            s.world = 1 + rng() % 9; s.rank = static_cast<int32_t>(rng() % 10) - 1; s.groups_total = static_cast<int32_t>(rng() % 520);
            s.m_max = static_cast<int32_t>(rng() % 300); s.n = static_cast<int32_t>(rng() % 5000); s.k = static_cast<int32_t>(rng() % 20000);
            s.chunks = static_cast<int32_t>(rng() % 6) - 1; s.max_tokens = (rng() & 1) ? static_cast<int32_t>(rng() % 100000) : 0;
            s.capacity_factor = (rng() & 1) ? 0.f : 0.25f * (rng() % 12); s.indexed = rng() & 1; s.policy = static_cast<int32_t>(rng() % 9) - 1;
            dga_sharded_layout_t l{};
            const int rc = dga_sharded_layout(&s, &l);
            int count = -1;
            const int rp = dga_sharded_plan(&s, nullptr, 0, &count);
            CHECK((rc == DGA_OK) == (rp == DGA_OK));
            if (rc != DGA_OK) continue;
            CHECK(count == l.steps && count > 0);
            std::vector<dga_sharded_step_t> steps(static_cast<size_t>(count));
            CHECK(dga_sharded_plan(&s, steps.data(), count, &count) == DGA_OK);
            CHECK(dga_sharded_plan(&s, steps.data(), count - 1, &count) == DGA_E_WORKSPACE);
            for (const auto &st : steps) {
                CHECK(st.stream >= 0 && st.stream <= 2);
                CHECK(st.event < l.events || st.event == -1);
                CHECK(st.row_begin >= 0 && st.row_begin + st.rows <= std::max<int64_t>(l.rows_total, 0) + (s.world == 1 ? 0 : 0));
                CHECK(st.group_begin >= 0 && st.group_begin + st.groups <= l.groups_local);
            }
        }
        CHECK(dga_sharded_layout(nullptr, nullptr) != DGA_OK);
        // the executor's argument checks return before any HIP call
        dga_sharded_shape_t s{};
        s.world = 2; s.rank = 0; s.groups_total = 8; s.m_max = 16; s.n = 128; s.k = 128; s.policy = -1;
        dga_sharded_buffers_t b{};
        void *streams[3] = {nullptr, nullptr, nullptr};
        CHECK(dga_sharded_forward(&s, &b, nullptr, nullptr, nullptr, 0, nullptr, 0, streams, nullptr, nullptr, nullptr) == DGA_E_NULL);
        CHECK(dga_sharded_forward(&s, &b, nullptr, nullptr, nullptr, -1, nullptr, 0, streams, nullptr, nullptr, nullptr) == DGA_E_RANGE);
        CHECK(dga_sharded_events_create(-1, nullptr) != DGA_OK);
        CHECK(dga_sharded_events_destroy(0, nullptr) == DGA_OK);
    }

    // ---- launch entry points: the checks in front of the first HIP call
    {
        dga_problem_t p = problem(128, 128, 128);
        dga_tiling_t t{};
        CHECK(dga_tiling(&p, &t) == DGA_OK);
        CHECK(dga_gemm_fp8_fp8_bf16_nt(nullptr, nullptr, nullptr, nullptr, nullptr, 128, 128, 128, &t, nullptr, 0, nullptr) != DGA_OK);
        CHECK(dga_gemm_fp8_fp8_bf16_nt(nullptr, nullptr, nullptr, nullptr, nullptr, -1, 128, 128, &t, nullptr, 0, nullptr) != DGA_OK);
        CHECK(dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 4, 16, 128, 128, 0, &t, nullptr, 0, nullptr) != DGA_OK);
        CHECK(dga_copy_rows(nullptr, 0, nullptr, nullptr, 0, nullptr, 16, 4, nullptr) != DGA_OK);
        CHECK(dga_copy_rows(nullptr, 0, nullptr, nullptr, 0, nullptr, -1, 4, nullptr) != DGA_OK);
        CHECK(dga_cast_to_fp8_1x128(nullptr, DGA_DT_BF16, 4, 128, nullptr, nullptr, nullptr) != DGA_OK);
        CHECK(dga_catlass_dynamic_matmul(nullptr, nullptr, nullptr, 4, 4, 4, DGA_DT_FP16, nullptr, 0, nullptr) != DGA_OK);
    }

    // ---- the cache and the selector from four threads at once
    {
        CHECK(dga_tiling_cache_open(nullptr) == DGA_OK);
        std::vector<std::thread> th;
        for (unsigned i = 0; i < 4; ++i) th.emplace_back([i] { tiling_sweep(100 + (i & 1), 1500); });
        th.emplace_back([] {   // ... while a fifth reloads the predictor and reopens / clears the cache under them
            const std::string f = tmp("dga_san_cache_mt.csv");
            for (int r = 0; r < 40; ++r) {
                (void)dga_predictor_load(nullptr);
                (void)dga_tiling_cache_open(r & 1 ? f.c_str() : nullptr);
                dga_predictor_unload();
                (void)dga_tiling_cache_clear();
            }
            (void)dga_tiling_cache_open(nullptr);
            std::remove(f.c_str());
        });
        for (auto &x : th) x.join();
        CHECK(dga_tiling_cache_clear() == DGA_OK);
    }
    std::printf("%s (%d failed checks)\n", fails ? "FAILED" : "ok", fails);
    return fails ? 1 : 0;
}
