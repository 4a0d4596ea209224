"""The predictor's selection strategies (dga_select_tiling_strategy / dga_select_kernel_with_predictor_ex) against the reference's own
select_tiling_strategy (/root/reference/get_best_config/get_best_config.py:431-525): tests/golden/select_strategy_vectors.json holds the
picks of that function, imported and called by tests/golden/make_golden.py on seeded candidate lists.
  greedy / topk_median  the same index;
  topk_dbscan           the same winning cluster (the reference returns a random member of it; this library its fastest member, or
                        member random_state % size), or the same fallback index where no cluster forms."""
import json
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
CASES = json.loads((ROOT / "tests" / "golden" / "select_strategy_vectors.json").read_text())["cases"]


def test_every_golden_pick(dga):
    assert len(CASES) > 250 and sum(1 for c in CASES if c["cluster"]) > 40
    for i, c in enumerate(CASES):
        preds = np.array(c["preds"], np.float32)
        picked, members = dga.select_tiling_strategy(preds, c["tiles"], method=c["method"], topk=c["topk"], dbscan_eps=c["eps"],
                                                     dbscan_min_samples=c["min_samples"])
        if c["cluster"] is None:
            assert picked == c["index"], (i, c["method"], c["topk"], picked, c["index"])
            assert members == []
        else:
            assert sorted(members) == sorted(c["cluster"]), (i, members, c["cluster"])
            assert picked == min(c["cluster"], key=lambda j: preds[j])          # the cluster's fastest member
            j, _ = dga.select_tiling_strategy(preds, c["tiles"], method=c["method"], topk=c["topk"], dbscan_eps=c["eps"],
                                              dbscan_min_samples=c["min_samples"], random_state=5)
            assert j == sorted(c["cluster"], key=lambda q: preds[q])[5 % len(c["cluster"])]


def test_argument_checks(dga):
    with pytest.raises(dga.DGAError):
        dga.select_tiling_strategy([1.0, 2.0], [[16, 16, 64], [32, 32, 64]], method="median_of_medians")
    with pytest.raises(dga.DGAError):
        dga.select_tiling_strategy([], [], method="greedy")
    assert dga.select_tiling_strategy([3.0], [[16, 16, 64]], method="topk_dbscan")[0] == 0


@pytest.mark.parametrize("method", ["greedy", "topk_median", "topk_dbscan"])
def test_select_kernel_with_predictor_takes_the_strategy(dga, method):
    """Whatever the strategy picks (or the fallback to the native tiling) is a tiling of the compiled menu."""
    for m, n, k in ((4096, 4096, 4096), (512, 4096, 7168), (64, 7168, 18432), (1300, 5000, 7680)):
        t, pred, native = dga.select_kernel_with_predictor(m, n, k, method=method)
        assert t.m1 in (16, 32, 64, 128, 256) and t.n1 in (128, 256) and t.splitkFactor >= 1
    t0, _, _ = dga.select_kernel_with_predictor(4096, 4096, 4096)
    t1, _, _ = dga.select_kernel_with_predictor(4096, 4096, 4096, method="greedy", topk=10)
    assert bytes(t0) == bytes(t1)
