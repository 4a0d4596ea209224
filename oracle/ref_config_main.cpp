// Driver (our code) around the REFERENCE's own header, compiled where it lies:
//   /root/reference/deep_gemm_ascend/framework/csrc/jit/get_best_config.hpp
// It prints the 28-int Config so that tests can pin our restatement
// (deepgemm_ascend_amd/csrc/dga_config.cpp) against the reference itself.
// TEST INFRASTRUCTURE ONLY; output binary goes to oracle/_ref/ (git-ignored).
//
// usage: ref_config best  batch m n k
//        ref_config bench m n k m_sections n_sections m_sec_o_blocks n_sec_o_blocks k_o_iter_blocks db_o_blocks
// prints the fields in the struct's declaration order (get_best_config.hpp:12-31).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "deep_gemm_ascend/framework/csrc/jit/get_best_config.hpp"

static void dump(const deep_gemm_ascend::Config &c)
{
    std::printf("%u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u %u\n",
                c.k_iters, c.batch, c.m, c.n, c.k, c.m_sections, c.n_sections, c.m_blocks, c.n_blocks,
                c.k_blocks, c.m_sc_blocks, c.n_sc_blocks, c.m_sec_o_blocks, c.n_sec_o_blocks,
                c.k_o_iter_blocks, c.db_o_blocks, c.m_o_fix, c.n_o_fix, c.k_o_fix, c.db_o_num, c.m_parts,
                c.n_parts, c.r_m_parts, c.r_n_parts, c.r_m_blocks, c.r_n_blocks, c.r_k_blocks, c.r_db_num);
}

int main(int argc, char **argv)
{
    if (argc == 6 && !std::strcmp(argv[1], "best")) {
        dump(deep_gemm_ascend::get_best_config(std::atoi(argv[2]), std::atoi(argv[3]), std::atoi(argv[4]),
                                               std::atoi(argv[5])));
        return 0;
    }
    if (argc == 11 && !std::strcmp(argv[1], "bench")) {
        unsigned v[9];
        for (int i = 0; i < 9; ++i) v[i] = (unsigned)std::atoi(argv[2 + i]);
        dump(deep_gemm_ascend::get_bench_config(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]));
        return 0;
    }
    std::fprintf(stderr, "usage: ref_config best batch m n k | bench m n k ms ns mso nso ko db\n");
    return 2;
}
