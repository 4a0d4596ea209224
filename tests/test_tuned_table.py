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
        bx = (int(r["dispatchPolicyTag"]) & 15) == 7    # a row of the bf16-exact policy's menu: a class of its own (one row per shape and class)
        key = (r["m"], r["n"], r["k"], r["groups"], r["contiguous"], bx)
        assert key not in seen, f"duplicate row {key}"
        seen.add(key)
        assert int(r["m1"]) in (16, 32, 64, 128, 256) and int(r["n1"]) in (128, 256) and int(r["stages"]) in (2, 3)
        assert int(r["kernelSerial"]) in ((0, 4, 5, 6, 7) if bx else (0, 1, 4, 5, 6)) and int(r["splitkFactor"]) >= 1   # (2, odd K in place, is never tabled)
        assert int(r["build"]) in ((0, 7, 8) if bx else (0,))
        if bx:
            assert (int(r["m1"]), int(r["n1"])) in ((128, 256), (128, 128), (64, 256), (64, 128), (32, 128), (16, 128))
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


def test_a_row_of_the_bf16_exact_class_is_what_a_default_call_runs(dga, tmp_path):
    """The operator's default arithmetic consults the cache first (select_kernel.cpp:371-378, cache.cpp:69-100): a row written with
    dispatchPolicyTag 7 (harness/sweep.py --arith bf16_exact) is the tiling of a default call, before the policy's cost model; the fast
    class of the same shape keeps its own row; the shipped table carries such rows (profiles/r06_bx_regret.txt)."""
    from deepgemm_ascend_amd import api
    path = tmp_path / "bx.csv"
    path.write_text("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,splitkFactor,stages,swizzleOffset,wavesM,wavesN,"
                    "dispatchPolicyTag,groups,contiguous,build\n"
                    "640,4096,7168,64,256,128,4,0,0,0,320,2,3,2,0,0,7,1,0,0\n"          # a tile + split the cost model does not pick
                    "640,4096,7168,128,128,128,0,0,0,0,160,1,3,2,2,2,4,1,0,0\n"         # the fast class's row of the same shape
                    "4096,4096,4096,128,256,128,0,0,0,0,512,1,3,8,0,0,7,1,0,8\n")        # names a build (the one-tile build)
    try:
        dga.tiling_cache_open(str(path))
        assert dga.tiling_cache_size() == 3
        t = dga.tiling(640, 4096, 7168, policy="bf16_exact")
        assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.swizzleOffset, t.dispatchPolicyTag, t.build) == (64, 256, 4, 2, 2, 7, 0)
        assert dga.tiling_check(t) == 0 and t.blockDim == 320
        f = dga.tiling(640, 4096, 7168)
        assert (f.m1, f.n1, f.dispatchPolicyTag) == (128, 128, 4)
        t = dga.tiling(4096, 4096, 4096, policy="bf16_exact")
        assert (t.m1, t.n1, t.build, t.dispatchPolicyTag) == (128, 256, 8, 7)
        # what a call without policy and tiling plans (the operator's default = bf16_exact) is the row
        p = api._planned(0, 640, 4096, 7168, 1, 0, False, False, None)
        assert (p.m1, p.n1, p.splitkFactor, p.dispatchPolicyTag) == (64, 256, 2, 7)
        # a shape without a row: the selector, as before
        s = dga.tiling(768, 4096, 7168, policy="bf16_exact")
        assert s.dispatchPolicyTag == 7 and dga.tiling_check(s) == 0
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()
    shipped = [r for r in _rows() if (int(r["dispatchPolicyTag"]) & 15) == 7]
    assert len(shipped) >= 4


def test_the_sweep_writes_decode_build_rows_the_cache_serves(dga, tmp_path):
    """harness/sweep.py --arith bf16_exact: the one-launch decode split-K (kernelSerial 6, build 10) is among the candidates of a short-M
    shape; a winner of that kind becomes a tag-7 row with its build, its split count and tiles x splits as blockDim, and the cache hands
    exactly that to a default call -- also where the selector's own rule would not name the build."""
    from deepgemm_ascend_amd.harness import sweep
    m, n, k = 64, 2112, 7168                         # 17 tiles: below the rule's 24
    assert dga.tiling(m, n, k, policy="bf16_exact").build == 0
    cands = [c for c in sweep.candidates_bx(m, n, k) if c.get("build") == 10]
    assert cands and all(c["m1"] == 64 and c["n1"] == 128 and c.get("wsk") and 1 <= c["splitk"] <= 8 for c in cands)
    assert {sweep.bx_serial(c) for c in cands} == {6}
    assert not [c for c in sweep.candidates_bx(1024, 4096, 7168) if c.get("build") == 10]      # more tiles than CUs: no such candidate
    win = max(cands, key=lambda c: c["splitk"])
    path = tmp_path / "bx_decode.csv"
    sweep.write_bx_rows(path, [((m, n, k), win)])
    row = path.read_text().strip().splitlines()[1].split(",")
    assert row[:5] == [str(m), str(n), str(k), "64", "128"] and row[6] == "6" and row[-1] == "10" and int(row[10]) == 17 * win["splitk"]
    try:
        dga.tiling_cache_open(str(path))
        t = dga.tiling(m, n, k, policy="bf16_exact")
        assert (t.m1, t.n1, t.kernelSerial, t.build, t.splitkFactor, t.blockDim) == (64, 128, 6, 10, win["splitk"], 17 * win["splitk"])
        assert dga.tiling_check(t) == 0 and dga.workspace_bytes(t) >= 17 * (win["splitk"] - 1) * (64 * 128 * 4 + 8)
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()
