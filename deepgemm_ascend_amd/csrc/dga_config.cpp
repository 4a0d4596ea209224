// The framework's 28-int tiling Config, restated.
// Reference: /root/reference/deep_gemm_ascend/framework/csrc/jit/get_best_config.hpp
//   struct Config :12-31, get_best_config :33-91, get_bench_config :93-153.
// One derivation serves both entry points (the reference duplicates it); the
// default knobs (1,1,3,8,20,10) are those hard-coded at :37-38,47-51.
// Pinned by tests/test_config.py against oracle/_ref/ref_config (the reference's own
// header compiled where it lies) and tests/golden/config_vectors.json.
#include <cstdint>
#include "dga_hip.h"

namespace {

struct Cfg28 {
    uint32_t k_iters, batch, m, n, k, m_sections, n_sections, m_blocks, n_blocks, k_blocks, m_sc_blocks,
        n_sc_blocks, m_sec_o_blocks, n_sec_o_blocks, k_o_iter_blocks, db_o_blocks, m_o_fix, n_o_fix, k_o_fix,
        db_o_num, m_parts, n_parts, r_m_parts, r_n_parts, r_m_blocks, r_n_blocks, r_k_blocks, r_db_num;
};
static_assert(sizeof(Cfg28) == 28 * 4, "Config is 28 uint32");

inline uint32_t up16(uint32_t x) { return (x + 15u) & ~15u; }
inline uint32_t cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }
inline uint32_t tail_or_full(uint32_t total, uint32_t step)
{
    const uint32_t r = total % step;
    return r ? r : step;
}

// Returns false when a knob is zero (the reference would divide by zero).
bool derive(uint32_t batch, uint32_t m, uint32_t n, uint32_t k, uint32_t ms, uint32_t ns, uint32_t mso,
            uint32_t nso, uint32_t kit, uint32_t db, Cfg28 &c)
{
    if (!ms || !ns || !mso || !nso || !kit || !db) return false;
    c.batch = batch; c.m = m; c.n = n; c.k = k;
    c.m_sections = ms; c.n_sections = ns;
    c.m_sec_o_blocks = mso; c.n_sec_o_blocks = nso; c.k_o_iter_blocks = kit; c.db_o_blocks = db;
    // 16-element "blocks" along each axis and the zero padding that implies
    c.m_blocks = up16(m) / 16; c.n_blocks = up16(n) / 16; c.k_blocks = up16(k) / 16;
    c.m_o_fix = up16(m) - m; c.n_o_fix = up16(n) - n; c.k_o_fix = up16(k) - k;
    c.db_o_num = kit / db;                       // L1->L0 sub-steps per K chunk
    c.r_m_blocks = tail_or_full(c.m_blocks, mso);  // last (possibly short) M / N part
    c.r_n_blocks = tail_or_full(c.n_blocks, nso);
    c.k_iters = cdiv(c.k_blocks, kit);
    const uint32_t k_tail = c.k_blocks % kit;
    if (k_tail == 0) {
        c.r_db_num = c.db_o_num;
        c.r_k_blocks = db;
    } else {
        c.r_db_num = cdiv(k_tail, db);
        c.r_k_blocks = k_tail - (c.r_db_num - 1) * db;
    }
    const uint32_t m_iters = cdiv(c.m_blocks, mso), n_iters = cdiv(c.n_blocks, nso);
    c.m_parts = m_iters / ms; c.n_parts = n_iters / ns;            // parts per core
    c.m_sc_blocks = c.m_parts * mso; c.n_sc_blocks = c.n_parts * nso;
    c.r_m_parts = m_iters - (ms - 1) * c.m_parts;                  // the last core takes the remainder
    c.r_n_parts = n_iters - (ns - 1) * c.n_parts;
    return true;
}

void flat(const Cfg28 &c, uint32_t out[28])
{
    const uint32_t *p = reinterpret_cast<const uint32_t *>(&c);
    for (int i = 0; i < 28; ++i) out[i] = p[i];
}

}  // namespace

extern "C" {

int dga_get_best_config(uint32_t batch, uint32_t m, uint32_t n, uint32_t k, uint32_t out[28])
{
    if (!out) return DGA_E_NULL;
    Cfg28 c{};
    derive(batch, m, n, k, 1, 1, 3, 8, 20, 10, c);
    flat(c, out);
    return DGA_OK;
}

int dga_get_bench_config(uint32_t m, uint32_t n, uint32_t k, uint32_t m_sections, uint32_t n_sections,
                         uint32_t m_sec_o_blocks, uint32_t n_sec_o_blocks, uint32_t k_o_iter_blocks,
                         uint32_t db_o_blocks, uint32_t out[28])
{
    if (!out) return DGA_E_NULL;
    Cfg28 c{};
    if (!derive(1, m, n, k, m_sections, n_sections, m_sec_o_blocks, n_sec_o_blocks, k_o_iter_blocks, db_o_blocks, c))
        return DGA_E_RANGE;
    flat(c, out);
    return DGA_OK;
}

// gemm_bench.hpp:68-81: params[6..27] = m,n,k,batch,k_iters,m_blocks,n_blocks,k_blocks,m_sc_blocks,n_sc_blocks,
// m_o_fix,n_o_fix,k_o_fix,db_o_num,m_parts,n_parts,r_m_parts,r_n_parts,r_m_blocks,r_n_blocks,r_k_blocks,r_db_num
int dga_bench_params_fill(uint32_t m, uint32_t n, uint32_t k, int32_t params[28])
{
    if (!params) return DGA_E_NULL;
    for (int i = 0; i < 6; ++i)
        if (params[i] <= 0) return DGA_E_RANGE;
    Cfg28 c{};
    derive(1, m, n, k, params[0], params[1], params[2], params[3], params[4], params[5], c);
    const uint32_t v[22] = {c.m, c.n, c.k, c.batch, c.k_iters, c.m_blocks, c.n_blocks, c.k_blocks, c.m_sc_blocks,
                            c.n_sc_blocks, c.m_o_fix, c.n_o_fix, c.k_o_fix, c.db_o_num, c.m_parts, c.n_parts,
                            c.r_m_parts, c.r_n_parts, c.r_m_blocks, c.r_n_blocks, c.r_k_blocks, c.r_db_num};
    for (int i = 0; i < 22; ++i) params[6 + i] = static_cast<int32_t>(v[i]);
    return DGA_OK;
}

// benchmark_util.h:78-85: m,n,k, the six knobs, batch, k_iters, ... r_db_num
int dga_bbit_params(uint32_t m, uint32_t n, uint32_t k, uint32_t m_sections, uint32_t n_sections,
                    uint32_t m_sec_o_blocks, uint32_t n_sec_o_blocks, uint32_t k_o_iter_blocks,
                    uint32_t db_o_blocks, uint32_t out[28])
{
    if (!out) return DGA_E_NULL;
    Cfg28 c{};
    if (!derive(1, m, n, k, m_sections, n_sections, m_sec_o_blocks, n_sec_o_blocks, k_o_iter_blocks, db_o_blocks, c))
        return DGA_E_RANGE;
    const uint32_t v[28] = {c.m, c.n, c.k, c.m_sections, c.n_sections, c.m_sec_o_blocks, c.n_sec_o_blocks,
                            c.k_o_iter_blocks, c.db_o_blocks, c.batch, c.k_iters, c.m_blocks, c.n_blocks,
                            c.k_blocks, c.m_sc_blocks, c.n_sc_blocks, c.m_o_fix, c.n_o_fix, c.k_o_fix, c.db_o_num,
                            c.m_parts, c.n_parts, c.r_m_parts, c.r_n_parts, c.r_m_blocks, c.r_n_blocks,
                            c.r_k_blocks, c.r_db_num};
    for (int i = 0; i < 28; ++i) out[i] = v[i];
    return DGA_OK;
}

}  // extern "C"
