"""CPU: operator hooks, the tiling restatement in reference mode (pinned by the reference's golden tuples),
the MI355X selection, and the CSV-backed tiling cache."""
import json
from pathlib import Path

import pytest

G = Path(__file__).parent / "golden"


def test_infer_shape_and_dtype(dga):
    assert dga.infer_shape([128, 64], [64, 256]) == (128, 256)      # catlass_dynamic_matmul.cpp:16-35
    with pytest.raises(dga.DGAError):
        dga.infer_shape([2, 128, 64], [64, 256])
    assert dga.infer_dtype(2, 2) == 2 and dga.infer_dtype(1, 1) == 1  # out = in (:37-46)
    assert dga.infer_dtype(3, 3) == 2                                  # fp8 in -> bf16 out
    with pytest.raises(dga.DGAError):
        dga.infer_dtype(1, 2)


def _tuple(t):
    return ([t.m1, t.n1, t.k1], t.kernelSerial, t.blockDim, [t.paddingTagA, t.paddingTagB, t.paddingTagC])


def test_reference_mode_matches_cpp_probes(dga):
    """C++ op_tiling outputs quoted in SURVEY.md 8(a7) (coreNum 24, NT)."""
    fx = json.loads((G / "op_tiling_vectors.json").read_text())
    for r in fx["survey_cpp_probes"]:
        t = dga.select_kernel(*r["shape"], platform=dga.platform_ascend910b(24))
        assert ([t.m1, t.n1, t.k1], t.kernelSerial, t.blockDim) == (r["tiling"], r["kernelSerial"], r["blockDim"])


def test_reference_mode_matches_python_mirror(dga):
    fx = json.loads((G / "op_tiling_vectors.json").read_text())
    n_alt = 0
    for r in fx["python_mirror"]:
        t = dga.select_kernel(*r["shape"], platform=dga.platform_ascend910b(r["coreNum"]))
        if r["python_only_alt_branch"]:
            # the mirror's extra 256x128 swap does not exist in select_kernel.cpp: we follow the C++
            n_alt += 1
            assert [t.m1, t.n1, t.k1] == r["do_tiling"], r
            continue
        assert _tuple(t) == (r["tiling"], r["kernelSerial"], r["blockDim"], r["padding"]), r
    assert n_alt <= 4


def test_reference_pinned_values(dga):
    # test_tiling_calculator.py:210-212: calculate(128,128,128) => SmallMatmul (kernelSerial 1)
    assert dga.select_kernel(128, 128, 128, platform=dga.platform_ascend910b(20)).kernelSerial == 1
    # TilingParams ctor (tiling_params.h:45-65): NT strides (k, k, n); swizzle 3 / (m > n ? 0 : 1)
    t = dga.select_kernel(4096, 2048, 7168, platform=dga.platform_ascend910b(24))
    assert (t.strideA, t.strideB, t.strideC) == (7168, 7168, 2048)
    assert t.swizzleOffset == 3 and t.swizzleDirection == 0 and t.splitkFactor == 1


def test_mi355x_selection_is_launchable(dga):
    pf = dga.platform_mi355x()
    assert (pf.coreNum, pf.l1Size, pf.xcdNum, pf.waveSize) == (256, 160 * 1024, 8, 64)
    menu = {(256, 256), (128, 256), (256, 128), (128, 128), (64, 256), (64, 128), (32, 256), (32, 128), (16, 256), (16, 128)}
    for shape in [(4096, 4096, 4096), (4096, 2048, 7168), (128, 2048, 7168), (8, 7168, 18432), (1, 128, 128),
                  (1279, 5003, 7696), (64, 24576, 1536), (5119, 6997, 9904)]:
        t = dga.select_kernel(*shape)
        assert (t.m1, t.n1) in menu and t.k1 == 128
        assert t.ldsBytes <= pf.l1Size and t.m1 * t.n1 * 4 <= pf.l0CSize
        tiles = -(-shape[0] // t.m1) * -(-shape[1] // t.n1)
        if t.kernelSerial == 6:   # workgroup split-K (decode rows): one workgroup per CU, each walking its share of the n-tiles
            assert shape[0] <= 16 and (t.m1, t.n1, t.splitkFactor, t.stages) == (16, 128, 1, 3)
            assert t.blockDim == min(-(-shape[1] // 16), pf.coreNum)
        elif t.kernelSerial == 5:   # whole waves of 256x256 tiles + the last partial wave in quarter tiles
            assert (t.m1, t.n1, t.splitkFactor) == (256, 256, 1) and 0 < tiles % 256 <= 64
            assert t.blockDim == tiles - tiles % 256 + 4 * (tiles % 256)
        else:
            assert t.blockDim == tiles * max(1, t.splitkFactor)
        assert (t.kernelSerial == 4) == (t.splitkFactor > 1)
        assert (t.paddingTagA, t.paddingTagB, t.paddingTagC) == (0, 0, 0)
        assert t.swizzleOffset >= 1
    t = dga.select_kernel(4096, 4096, 4096)
    assert (t.m1, t.n1, t.blockDim) == (256, 256, 256)   # one full wave of the 256 CUs
    tq = dga.select_kernel(1024, 18432, 7168)             # 288 tiles: one wave + 32 tiles in quarter tiles
    assert (tq.m1, tq.n1, tq.kernelSerial, tq.blockDim) == (256, 256, 5, 256 + 32 * 4)
    tg = dga.select_kernel(128, 2048, 7168, groups=256, expected_m=128)
    assert tg.groups == 256 and tg.m1 == 128 and tg.blockDim == 256 * (2048 // tg.n1)
    assert dga.select_kernel(0, 128, 128).blockDim == 0
    # decode rows: the one-launch workgroup split-K where it won its cold sweep (M <= 16, N <= 10240, 2048 <= K <= 18432)
    assert dga.select_kernel(8, 7168, 18432).kernelSerial == 6 and dga.select_kernel(16, 4096, 7168).kernelSerial == 6
    assert dga.select_kernel(32, 4096, 7168).kernelSerial != 6 and dga.select_kernel(8, 18432, 7168).kernelSerial == 6
    assert dga.select_kernel(8, 129280, 7168).kernelSerial != 6 and dga.select_kernel(8, 24576, 1536).kernelSerial != 6
    assert dga.select_kernel(8, 57344, 8192).kernelSerial == 6 and dga.select_kernel(4, 3584, 18944).kernelSerial == 6   # (round 4's fourth grid)
    assert dga.select_kernel(8, 5120, 27648).kernelSerial != 6                                                            # (216 k blocks: a slice too long)
    assert dga.select_kernel(8, 7168, 1536).kernelSerial != 6


def test_non_nt_layout_rejected(dga):
    import ctypes
    from deepgemm_ascend_amd import _lib
    p = _lib.Problem(64, 64, 64, 1, 0, 0, 0, 0, _lib.DT_FP8_E4M3FN)  # B row-major: not the op's layout
    t = _lib.Tiling()
    assert _lib.lib().dga_select_kernel(ctypes.byref(p), None, ctypes.byref(t)) == -2


def test_tiling_cache_csv_roundtrip(dga, tmp_path):
    """cache.cpp:22-101 / csv.cpp:31-140; literal fixture rows from the reference's csv_test.cpp:33-35."""
    path = tmp_path / "cache.csv"
    path.write_text("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim\n"
                    "512,512,512,128,256,256,0,0,0,0,24\n"
                    "1024,1024,1024,256,256,256,1,1,0,0,24\n")
    try:
        dga.tiling_cache_open(str(path))
        assert dga.tiling_cache_size() == 2
        t = dga.tiling(512, 512, 512)                       # hit: values come from the file
        # (k1: the reference's L1 k tile; the kernels here step one 128-wide scale block whatever it says -- the cached row is
        #  normalised onto the menu so that dga_tiling_check accepts what dga_tiling returns)
        assert (t.m1, t.n1, t.k1, t.kernelSerial, t.blockDim) == (128, 256, 128, 0, 24)
        assert dga.tiling_check(t) == 0
        t = dga.tiling(1024, 1024, 1024)
        assert (t.m1, t.n1, t.kernelSerial, t.paddingTagA) == (256, 256, 1, 1)
        t3 = dga.tiling(2048, 2048, 2048)                   # miss: computed, appended
        assert dga.tiling_cache_size() == 3
        rows = path.read_text().strip().splitlines()
        assert len(rows) == 4 and rows[-1].startswith("2048,2048,2048,")
        assert rows[-1].split(",")[3:6] == [str(t3.m1), str(t3.n1), str(t3.k1)]
        again = dga.tiling(2048, 2048, 2048)                # second call: same tile, no new row
        assert (again.m1, again.n1) == (t3.m1, t3.n1)
        assert len(path.read_text().strip().splitlines()) == 4
        # reopen: the appended row is read back
        dga.tiling_cache_open(str(path))
        assert dga.tiling_cache_size() == 3
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


def test_tiling_cache_creates_header_for_new_file(dga, tmp_path):
    path = tmp_path / "new.csv"
    try:
        dga.tiling_cache_open(str(path))
        # the reference's eleven columns (csv.cpp:23-26) first, the CDNA4 columns behind them
        assert path.read_text() == ("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,"
                                    "splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag,groups,contiguous,build\n")
        dga.tiling(256, 256, 256)
        assert len(path.read_text().strip().splitlines()) == 2
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


@pytest.mark.parametrize("m,n,k", [(64, 7168, 18432), (8, 18432, 7168), (1024, 18432, 7168), (4096, 4096, 4096),
                                   (512, 4096, 7168), (4096, 2048, 7168)])
def test_tiling_round_trips_through_a_new_cache_file(dga, tmp_path, m, n, k):
    """tiling -> CSV row -> a fresh cache (= the next process) -> the identical dga_tiling_t: split-K factor, stages,
    wave grid, raster and schedule survive a restart, so the same shape runs the same kernel."""
    path = tmp_path / "rt.csv"
    try:
        dga.tiling_cache_open(str(path))
        first = dga.tiling(m, n, k).as_dict()
        dga.tiling_cache_open(str(path))          # drops the in-memory map, re-reads the file
        assert dga.tiling_cache_size() == 1
        again = dga.tiling(m, n, k).as_dict()
        assert again == first
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


@pytest.mark.parametrize("kw", [dict(m=128, n=2048, k=7168, groups=256, expected_m=128),
                                dict(m=4096, n=4096, k=7168, groups=32, contiguous=True),
                                dict(m=8192, n=4096, k=7168, groups=8, contiguous=True)])
def test_grouped_tilings_round_trip_too(dga, tmp_path, kw):
    """The groups / contiguous columns key masked-grouped and contiguous-grouped problems in the same file: the loader-wave
    policy, the 4-wave build of the weight stream and the two-pass 256x256 tiling of long groups survive a restart, and a
    dense problem of the same (m, n, k) stays a different entry."""
    path = tmp_path / "grp.csv"
    try:
        dga.tiling_cache_open(str(path))
        first = dga.tiling(**kw).as_dict()
        dense = dga.tiling(kw["m"], kw["n"], kw["k"]).as_dict()
        dga.tiling_cache_open(str(path))
        assert dga.tiling_cache_size() == 2
        assert dga.tiling(**kw).as_dict() == first
        assert dga.tiling(kw["m"], kw["n"], kw["k"]).as_dict() == dense
        assert first["groups"] == kw["groups"] and dense["groups"] == 1
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


def test_persistent_form_where_the_raster_exceeds_the_cus(dga):
    """A loader-wave tiling whose raster holds more tiles than the chip has CUs takes the persistent build
    (dispatchPolicyTag 5); one tile per CU or fewer, split-K and the quarter-tile tail keep the one-tile builds."""
    big = dga.tiling(128, 2048, 7168, groups=256, expected_m=128)          # 2048 tiles of 128x256
    assert (big.m1, big.n1, big.stages, big.dispatchPolicyTag) == (128, 256, 3, dga.api.POLICY_PERSISTENT)
    one = dga.select_kernel(4096, 2048, 7168)                               # BASELINE configs[2]: 256 tiles, one per CU
    assert (one.m1, one.n1, one.dispatchPolicyTag) == (128, 256, dga.api.POLICY_LOADER_WAVES)
    few = dga.select_kernel(128, 2048, 7168, groups=16, expected_m=128)    # 128 tiles, but a weight stream: the persistent
    assert few.dispatchPolicyTag == dga.api.POLICY_PERSISTENT              # build moves it with the non-temporal policy
    dense_few = dga.select_kernel(2048, 2048, 7168)                         # a dense raster of 128 loader-wave tiles
    assert dense_few.dispatchPolicyTag in (dga.api.POLICY_LOADER_WAVES, dga.api.POLICY_PLAIN, dga.api.POLICY_CONTINUOUS)
    for (m, n, k) in [(64, 7168, 18432), (128, 4096, 7168), (8, 18432, 7168)]:
        t = dga.select_kernel(m, n, k)
        assert t.dispatchPolicyTag != dga.api.POLICY_PERSISTENT or t.splitkFactor == 1


def test_dense_selector_follows_the_device_timed_sweep(dga):
    """The dense selector picks (tile, split-K) together by the cost model fitted to the device-timed sweep
    (scripts/fit_heuristic.py, profiles/r03_predictor): tall tiles with a split where the tile-first rule took 16-row
    tiles without one; three LDS stages wherever the tile has such a build; no loader waves on 128x256 under split-K; the
    persistent form only on the tall tiles; a weight stream of M <= 64 rows is not cut along M (every tile row streams B again)."""
    for (m, n, k) in [(310, 1280, 15744), (992, 512, 11776), (331, 1536, 11904)]:      # 2.5-3.5x off the best before
        t = dga.select_kernel(m, n, k)
        assert t.m1 >= 64 and t.splitkFactor >= 3 and t.stages == 3, (m, n, k, t.m1, t.n1, t.splitkFactor)
    t = dga.select_kernel(630, 8320, 14080)
    assert (t.m1, t.n1) == (128, 256) and t.splitkFactor > 1 and t.dispatchPolicyTag == dga.api.POLICY_PLAIN and (t.wavesM, t.wavesN) == (2, 4)
    t = dga.select_kernel(1536, 16384, 7168)                                           # 768 tiles of 128x256: persistent
    assert (t.m1, t.n1, t.dispatchPolicyTag) == (128, 256, dga.api.POLICY_PERSISTENT)
    for (m, n, k) in [(4658, 12032, 2304), (300, 30000, 512), (2000, 2000, 256)]:      # whatever is picked: short tiles never persistent
        t = dga.select_kernel(m, n, k)
        if (t.m1, t.n1) in ((64, 128), (16, 128), (32, 128)):
            assert t.dispatchPolicyTag != dga.api.POLICY_PERSISTENT
    for (m, n, k) in [(8, 18432, 7168), (16, 7296, 15232), (32, 9344, 15744), (64, 4096, 7168)]:
        t = dga.select_kernel(m, n, k)
        assert t.m1 >= m and t.m1 < 2 * max(m, 16) and t.stages == 3, (m, t.m1)       # one tile row, the smallest tile that covers M
    # the BASELINE headline shapes keep their builds
    assert [(t.m1, t.n1, t.splitkFactor) for t in (dga.select_kernel(4096, 4096, 4096), dga.select_kernel(4096, 2048, 7168))] == \
        [(256, 256, 1), (128, 256, 1)]


def test_persistent_continuous_form_on_full_tile_rasters(dga):
    """256x256 continuous tilings of dense problems made of full tiles take the persistent form (dispatchPolicyTag 6) where
    the raster holds more tiles than CUs; one tile per CU, edges, or the quarter-tile-free small cases keep policy 2."""
    big = dga.select_kernel(8192, 8192, 8192)
    assert (big.m1, big.n1, big.dispatchPolicyTag) == (256, 256, dga.api.POLICY_CONTINUOUS_PERSISTENT)
    one = dga.select_kernel(4096, 4096, 4096)                              # BASELINE configs[1]: one tile per CU
    assert (one.m1, one.n1, one.dispatchPolicyTag) == (256, 256, dga.api.POLICY_CONTINUOUS)
    edge = dga.select_kernel(8192 + 8, 8192, 4096)                          # an M edge: not a raster of full tiles
    assert edge.dispatchPolicyTag != dga.api.POLICY_CONTINUOUS_PERSISTENT


def test_reference_format_file_keeps_its_format(dga, tmp_path):
    """A file that carries only the reference's eleven columns is appended to in that format."""
    path = tmp_path / "ref.csv"
    path.write_text("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim\n")
    try:
        dga.tiling_cache_open(str(path))
        dga.tiling(64, 7168, 18432)
        rows = path.read_text().strip().splitlines()
        assert len(rows) == 2 and len(rows[1].split(",")) == 11
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


def test_workspace_bytes(dga):
    assert dga.workspace_bytes(dga.select_kernel(4096, 4096, 4096)) == 0


def test_tiling_cache_reads_swept_cdna4_columns(dga, tmp_path):
    """A sweep's CSV carries the reference's 11 columns plus CDNA4 columns; a hit then reproduces the swept build."""
    path = tmp_path / "swept.csv"
    path.write_text("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,"
                    "splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag\n"
                    "64,4096,7168,64,256,128,4,0,0,0,64,4,3,1,0,0,0\n"
                    "4096,4096,4096,256,256,128,0,0,0,0,256,1,2,16,0,0,1\n")
    try:
        dga.tiling_cache_open(str(path))
        t = dga.tiling(64, 4096, 7168)
        assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.stages, t.swizzleOffset) == (64, 256, 4, 4, 3, 1)
        assert dga.workspace_bytes(t) >= 4 * 64 * 4096 * 4
        t = dga.tiling(4096, 4096, 4096)
        assert (t.m1, t.n1, t.dispatchPolicyTag, t.swizzleOffset, t.splitkFactor) == (256, 256, 1, 16, 1)
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


@pytest.mark.parametrize("m,n,k,g", [(4096, 2048, 7168, 32), (16384, 7168, 2048, 128), (384, 4096, 7168, 3), (128, 128, 128, 1)])
def test_contiguous_layout_tiles_divide_the_alignment(dga, m, n, k, g):
    """Contiguous-grouped layout, short groups: tile height <= 128 and a divisor of it, no split-K, groups = number of
    B matrices."""
    t = dga.tiling(m, n, k, groups=g, contiguous=True)
    assert t.contiguous == 1 and t.groups == g
    assert 0 < t.m1 <= 128 and 128 % t.m1 == 0
    assert t.splitkFactor == 1
    assert t.blockDim == -(-m // t.m1) * -(-n // t.n1)
    # the dense tiling of the same (m,n,k) is unaffected (the cache is not shared)
    assert dga.tiling(m, n, k).contiguous == 0


@pytest.mark.parametrize("m,n,k,g", [(8192, 4096, 7168, 8), (32768, 7168, 2048, 4)])
def test_contiguous_layout_long_groups_take_two_pass_tiles(dga, m, n, k, g):
    """>= 512 rows per group on average: the 256x256 tile with a doubled grid (pass 1 covers straddling tiles)."""
    t = dga.tiling(m, n, k, groups=g, contiguous=True)
    assert (t.m1, t.n1, t.contiguous) == (256, 256, 1)
    assert t.blockDim == 2 * -(-m // 256) * -(-n // 256)


def test_swept_row_keeps_the_build_the_sweep_timed(dga, tmp_path):
    """A swept row whose file carries the dispatchPolicyTag column names the build that was timed: a 3-stage tile on the
    plain loop (policy 0) stays there.  The same row in a file from before that column existed is upgraded to the tile's
    loader-wave build, as every heuristic pick is."""
    head = "m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,splitkFactor,stages,swizzleOffset,wavesM,wavesN"
    row = "2048,2048,7168,128,256,128,0,0,0,0,128,1,3,4,2,4"
    try:
        new = tmp_path / "with_policy.csv"
        new.write_text(head + ",dispatchPolicyTag\n" + row + ",0\n")
        dga.tiling_cache_open(str(new))
        t = dga.tiling(2048, 2048, 7168)
        assert (t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.dispatchPolicyTag) == (128, 256, 3, 2, 4, dga.api.POLICY_PLAIN)
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


def test_contiguous_rows_share_a_bucketed_key(dga, tmp_path):
    """Prefill serving changes the contiguous layout's row count on almost every call: the cache keys it by 128 x a power
    of two, so the map and the file hold a handful of rows per (n, k, groups) instead of one per call."""
    path = tmp_path / "contig.csv"
    try:
        dga.tiling_cache_open(str(path))
        for msum in (4096 + 128, 4096 + 256, 6144, 8192 - 128, 8192):
            t = dga.tiling(msum, 4096, 7168, groups=8, contiguous=True)
            assert t.m == msum and t.m1 in (128, 256)
            assert t.blockDim == (-(-msum // t.m1)) * (4096 // t.n1) * (2 if t.m1 > 128 else 1)
        assert dga.tiling_cache_size() == 1
        assert len(path.read_text().strip().splitlines()) == 2      # header + one row
        dga.tiling(8192 + 128, 4096, 7168, groups=8, contiguous=True)
        assert dga.tiling_cache_size() == 2
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()


def test_bf16_exact_policy_has_its_own_tiling(dga):
    """dga_tiling_bf16_exact: dispatchPolicyTag 7, a tile of THAT policy's menu, picked by the cost model fitted to that menu's
    own device-timed sweep (profiles/r03_predictor/bf16_exact_fit.txt); grouped layouts keep the fast
    tiling's tile; nothing is written to the tiling cache under the policy's name."""
    menu = {(128, 256), (128, 128), (64, 256), (64, 128), (32, 128)}
    for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (1024, 4096, 7168), (512, 4096, 7168), (128, 4096, 7168),
                      (64, 7168, 18432), (8, 18432, 7168), (300, 200, 256), (1279, 5003, 7681), (128, 7168, 18432), (64, 2112, 7168)]:
        t = dga.tiling(m, n, k, policy="bf16_exact")
        assert t.dispatchPolicyTag == dga.api.POLICY_BF16_EXACT and (t.m, t.n, t.k) == (m, n, k)
        if t.kernelSerial == 6 and t.build == 10:   # a few 64-row tiles: the one-launch split-K of the 64 x 128 tile (DGA_BUILD_BX_DECODE)
            assert 17 <= m <= 512 and (t.m1, t.n1, t.stages) == (64, 128, 3) and 1 <= t.splitkFactor <= 8
            assert t.blockDim == -(-m // 64) * -(-n // 128) * t.splitkFactor <= 256
            assert dga.workspace_bytes(t) >= (-(-m // 64) * -(-n // 128) * (t.splitkFactor - 1) * (64 * 128 * 4 + 8) if t.splitkFactor > 1 else 0)
            continue
        if t.kernelSerial == 6:   # decode rows: the workgroup split-K with this policy's arithmetic
            assert m <= 32 and (t.m1, t.n1, t.splitkFactor, t.stages) == (16 if m <= 16 else 32, 128, 1, 3)
            assert t.blockDim == min(-(-n // 16), 256)
        elif k % 16 == 0:
            assert (t.m1, t.n1) in menu, (m, n, k, t.m1, t.n1)
            assert t.blockDim == -(-m // t.m1) * -(-n // t.n1) * t.splitkFactor
        assert (t.splitkFactor > 1) == (t.kernelSerial == 4)
        assert dga.workspace_bytes(t) >= (m * n * 4 * t.splitkFactor if t.splitkFactor > 1 else 0)
    assert (dga.tiling(4096, 4096, 4096, policy="bf16_exact").m1, dga.tiling(4096, 4096, 4096, policy="bf16_exact").n1) == (128, 256)
    mid = dga.tiling(1024, 4096, 7168, policy="bf16_exact")   # 256 tiles of 128x128, one per CU: 57 us against 85 for 128x256 (r03_vgpr_form.txt)
    assert (mid.m1, mid.n1) == (128, 128) and mid.splitkFactor == 1
    assert dga.tiling(8, 18432, 7168, policy="bf16_exact").kernelSerial == 6 and dga.tiling(32, 4096, 7168, policy="bf16_exact").kernelSerial == 6
    assert dga.tiling(128, 4096, 7168, policy="bf16_exact").build == 10 and dga.tiling(128, 7168, 18432, policy="bf16_exact").build == 0
    assert dga.tiling(48, 18432, 7168, policy="bf16_exact").m1 in (32, 64) and dga.tiling(32, 18432, 7168, policy="bf16_exact").kernelSerial != 6
    g = dga.tiling(128, 2048, 7168, groups=256, expected_m=128, policy="bf16_exact")
    f = dga.tiling(128, 2048, 7168, groups=256, expected_m=128)
    assert g.dispatchPolicyTag == dga.api.POLICY_BF16_EXACT and (g.m1, g.n1) == (f.m1, f.n1) == (128, 256)
    assert (g.wavesM, g.wavesN) != (2, 2) and g.stages == 3 and g.build == 9   # the masked grouped layout's own kernel (DGA_BUILD_BX_GROUPED)
    for hint in (4, 16, 64):   # ... whatever the hint says: the kernel skips the rows that do not exist itself
        h = dga.tiling(128, 2048, 7168, groups=256, expected_m=hint, policy="bf16_exact")
        assert (h.m1, h.n1, h.build) == (128, 256, 9)
    small = dga.tiling(64, 2048, 7168, groups=256, expected_m=16, policy="bf16_exact")   # experts of at most 64 rows keep the fast tiling's tile
    assert small.build == 0 and small.m1 <= 64
    assert dga.tiling(4096, 4096, 4096).dispatchPolicyTag != dga.api.POLICY_BF16_EXACT      # the cache entry is the fast path's
    assert dga.tiling(4096, 4096, 4096, policy="strict").dispatchPolicyTag == dga.api.POLICY_STRICT
