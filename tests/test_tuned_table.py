"""The preloaded tiling table (deepgemm_ascend_amd/tuned/mi355x.csv; the counterpart of the reference's CSV tiling cache,
/root/reference/aclnn_catlass_dynamic_matmul/op_host/op_tiling/cache.cpp:22-101, select_kernel.cpp:371-378) and its decode-grid
fallback: a dense problem of M <= 128 rows that misses the table takes the swept row of the same (N, K) at the next row count of
the grid the cold sweep covered (profiles/r04_sweep_decode)."""
import csv
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
TABLE = ROOT / "deepgemm_ascend_amd" / "tuned" / "mi355x.csv"


def _rows():
    with open(TABLE) as f:
        return list(csv.DictReader(f))


@pytest.fixture
def table(dga, tmp_path):
    """The shipped table loaded from a scratch copy (other tests of the process open / clear the cache; a cache opened from a path
    appends its misses to that file, which must not be the shipped one)."""
    import shutil
    copy = tmp_path / "mi355x.csv"
    shutil.copy(TABLE, copy)
    dga.tiling_cache_open(str(copy))
    yield dga
    dga.tiling_cache_open(None)


def test_table_is_well_formed(dga):
    rows = _rows()
    assert len(rows) >= 100
    seen = set()
    for r in rows:
        key = (r["m"], r["n"], r["k"], r["groups"], r["contiguous"])
        assert key not in seen, f"duplicate row {key}"
        seen.add(key)
        assert int(r["m1"]) in (16, 32, 64, 128, 256) and int(r["n1"]) in (128, 256) and int(r["stages"]) in (2, 3)
        assert int(r["kernelSerial"]) in (0, 1, 4, 5, 6) and int(r["splitkFactor"]) >= 1   # (2, odd K in place, is never tabled)
        if int(r["kernelSerial"]) == 6:   # the workgroup split-K on LDS-DMA rings: decode rows only (profiles/r04_sweep_wskd)
            assert int(r["m"]) <= 32 and int(r["stages"]) == 3 and int(r["splitkFactor"]) == 1 and r["groups"] == "1"


@pytest.mark.parametrize("m,bucket", [(24, 32), (17, 32), (50, 64), (100, 128), (5, 8), (2, 4), (128, 128)])
def test_decode_rows_fall_back_to_the_next_swept_row_count(table, m, bucket):
    dga = table
    n, k = 18432, 7168
    want = next(r for r in _rows() if (int(r["m"]), int(r["n"]), int(r["k"]), int(r["groups"])) == (bucket, n, k, 1))
    t = dga.tiling(m, n, k)
    assert (t.m1, t.n1, t.splitkFactor, t.stages, t.kernelSerial) == tuple(int(want[c]) for c in ("m1", "n1", "splitkFactor", "stages", "kernelSerial"))
    assert t.m1 >= min(m, 128) or t.m1 * ((m + t.m1 - 1) // t.m1) >= m
    assert t.blockDim == ((m + t.m1 - 1) // t.m1) * ((n + t.n1 - 1) // t.n1) * t.splitkFactor


def test_no_fallback_beyond_the_grid_or_off_its_shapes(table):
    dga = table
    t_sel = dga.tiling(24, 5120, 3328)        # an (N, K) the sweep never saw: the fitted selector's pick, whatever it is
    assert t_sel.m1 in (16, 32, 64) and t_sel.n1 in (128, 256)
    t_big = dga.tiling(129, 18432, 7168)      # M > 128: not a decode row
    assert t_big.m1 >= 128
