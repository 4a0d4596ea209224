"""CPU: the 28-int Config restatement against the reference's own header (oracle/_ref/ref_config when
/root/reference is present) and against the committed vectors made from it."""
import json
from pathlib import Path

import pytest

G = Path(__file__).parent / "golden"


def test_config_vectors(dga):
    fx = json.loads((G / "config_vectors.json").read_text())
    for row in fx["best"]:
        assert dga.get_best_config(*row["args"]) == row["config"], row["args"]
    for row in fx["bench"]:
        assert dga.get_bench_config(*row["args"]) == row["config"], row["args"]


def test_survey_probe_values(dga):
    """C2 defaults => k_iters=13, parts=(86,32), r_blocks=(1,8,6), r_db_num=2."""
    c = dict(zip(dga.CONFIG_FIELDS, dga.get_best_config(1, 4096, 4096, 4096)))
    assert (c["k_iters"], c["m_parts"], c["n_parts"], c["r_m_blocks"], c["r_n_blocks"], c["r_k_blocks"], c["r_db_num"]) \
        == (13, 86, 32, 1, 8, 6, 2)
    c = dict(zip(dga.CONFIG_FIELDS, dga.get_bench_config(96, 1536, 5952, 1, 1, 3, 8, 20, 10)))
    assert (c["k_iters"], c["m_parts"], c["n_parts"], c["r_m_blocks"], c["r_n_blocks"], c["r_k_blocks"], c["r_db_num"]) \
        == (19, 2, 12, 3, 8, 2, 2)


def test_against_reference_binary_when_present(dga, oracle):
    import random
    try:
        oracle.ref_config("best", 1, 16, 16, 16)
    except FileNotFoundError:
        pytest.skip("oracle/_ref/ref_config not built (reference tree absent)")
    rnd = random.Random(7)
    for _ in range(60):
        m, n, k = rnd.randint(1, 9000), rnd.randint(1, 9000), rnd.randint(1, 20000)
        assert dga.get_best_config(rnd.randint(1, 4), m, n, k)[2:] == oracle.ref_config("best", 1, m, n, k)[2:]
        knobs = [rnd.randint(1, 6), rnd.randint(1, 6), rnd.randint(1, 16), rnd.randint(1, 16)]
        db = rnd.randint(1, 12)
        knobs += [db * rnd.randint(1, 4), db]
        assert dga.get_bench_config(m, n, k, *knobs) == oracle.ref_config("bench", m, n, k, *knobs)


def test_param_orders(dga):
    """The two flat orders differ: python-binding order (gemm_bench.hpp:68-81) vs harness order
    (benchmark_util.h:78-85)."""
    knobs = (2, 3, 4, 8, 20, 10)
    c = dict(zip(dga.CONFIG_FIELDS, dga.get_bench_config(300, 500, 700, *knobs)))
    filled = dga.bench_params_fill(300, 500, 700, knobs)
    assert filled[:6] == list(knobs)
    names = "m n k batch k_iters m_blocks n_blocks k_blocks m_sc_blocks n_sc_blocks m_o_fix n_o_fix k_o_fix db_o_num " \
            "m_parts n_parts r_m_parts r_n_parts r_m_blocks r_n_blocks r_k_blocks r_db_num".split()
    assert filled[6:] == [c[x] for x in names]
    bb = dga.bbit_params(300, 500, 700, *knobs)
    names2 = "m n k m_sections n_sections m_sec_o_blocks n_sec_o_blocks k_o_iter_blocks db_o_blocks batch".split() + names[4:]
    assert bb == [c[x] for x in names2]


def test_zero_knob_is_rejected(dga):
    with pytest.raises(dga.DGAError):
        dga.get_bench_config(16, 16, 16, 1, 1, 0, 8, 20, 10)
