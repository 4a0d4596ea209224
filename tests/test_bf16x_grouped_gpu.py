"""The bf16-exact policy's masked-grouped kernel (csrc/gemm_fp8_bf16x_grouped_kernel.hpp: persistent, two k blocks of the ring in
flight, the loop unrolled for the 0..4 m-tiles of a wave's rows that exist) against the one-tile build of the same tile: the same
arithmetic in the same order, so the bar is BIT IDENTITY; against the oracle it is the policy's bar (tests/test_bf16_exact_gpu.py);
rows at or beyond masked_m stay untouched.  Counterpart in the reference: its m_parts walk multiplies only the blocks that exist
(/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:189-200).  tiling.build = 9 (DGA_BUILD_BX_GROUPED, what
dga_tiling_bf16_exact names for this layout) is the kernel, 8 the one-tile build."""
import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev

pytestmark = pytest.mark.gpu


def _tiling(dga, mmax, n, k, groups, grouped_kernel, raster=1):
    t = dga.tiling(mmax, n, k, groups=groups, policy="bf16_exact")
    t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.dispatchPolicyTag = 128, 256, 1, 0, 7
    t.stages, t.build, t.wavesM, t.wavesN, t.swizzleOffset = 3, (9 if grouped_kernel else 8), 2, 4, raster
    return t


def _stack(oracle, g, mmax, n, k, seed):
    parts = [oracle.make_inputs(mmax, n, k, seed=seed + i) for i in range(4)]
    return tuple(np.stack([parts[i % 4][j] for i in range(g)]) for j in range(4))


def _both(dga, A, SFA, B, SFB, masked, fill=-7.0):
    g, mmax, _ = A.shape
    n = B.shape[1]
    k = A.shape[2]
    outs = []
    for grouped_kernel in (True, False):
        out = torch.full((g, mmax, n), fill, dtype=torch.bfloat16, device="cuda")
        t = _tiling(dga, mmax, n, k, g, grouped_kernel)
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((_dev(A), _dev(SFA)), (_dev(B), _dev(SFB)), out, _dev(masked),
                                                  expected_m=mmax, policy="bf16_exact", sync=True, tiling_=t)
        outs.append(_bits(out))
    return outs


# every count of m-tiles in either half of the tile (0..4 in rows 0..63, 0..4 in rows 64..127), edges of each, empty experts
ALL_L = np.array([0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 79, 80, 81, 95, 96, 97, 111, 112, 113, 127, 128, 5, 70, 100,
                  0, 128, 64, 1], np.int32)


@pytest.mark.parametrize("k", [256, 384, 128 * 9, 512 + 16])
def test_every_row_count_is_bit_identical_to_the_one_tile_build(dga, oracle, k):
    """32 experts x 8 n-tiles = 256 tiles (one per workgroup) and 48 experts (384 tiles: half the workgroups walk two tiles, so tiles of
    different m-tile counts follow each other in one workgroup)."""
    for g in (32, 48):
        n, mmax = 2048, 128
        A, SFA, B, SFB = _stack(oracle, g, mmax, n, k, seed=10 + k)
        masked = np.resize(ALL_L, g).astype(np.int32)
        if g == 48:
            masked = masked[::-1].copy()
        got, ref = _both(dga, A, SFA, B, SFB, masked)
        assert np.array_equal(got, ref), f"g {g}: {int((got != ref).sum())} of {got.size} outputs differ"
        init = _bits(torch.tensor([-7.0], dtype=torch.bfloat16))[0]
        for i in range(g):
            assert (got[i, int(masked[i]):] == init).all(), "rows >= masked_m were written"
        for i in (3, 9, 17, 26):
            mm = int(masked[i])
            if mm and k <= 384:
                want = oracle.gemm_fp8_fp8_bf16_nt(A[i, :mm], SFA[i, :mm], B[i], SFB[i], threads=8)
                _assert_bar(oracle, got[i, :mm], want, A[i, :mm], SFA[i, :mm], B[i], SFB[i])


def test_long_walks_and_ragged_n(dga, oracle):
    """600 experts x 3 n-tiles (N = 640: the last tile is half columns, so the scalar store path and the counted wait's fallback run)
    on 256 workgroups: every workgroup walks seven tiles."""
    g, mmax, n, k = 600, 128, 640, 256
    A, SFA, B, SFB = _stack(oracle, g, mmax, n, k, seed=77)
    rng = np.random.default_rng(5)
    masked = rng.integers(0, mmax + 1, size=g).astype(np.int32)
    got, ref = _both(dga, A, SFA, B, SFB, masked)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ"


def test_small_m_max_and_two_tile_rows(dga, oracle):
    """m_max below and above the tile height: 96 (one tile row, the second half of the waves has at most two m-tiles) and 200 (two tile
    rows per expert, the second one 72 rows)."""
    for mmax in (96, 200):
        g, n, k = 24, 1024, 384
        A, SFA, B, SFB = _stack(oracle, g, mmax, n, k, seed=5 + mmax)
        rng = np.random.default_rng(mmax)
        masked = rng.integers(0, mmax + 1, size=g).astype(np.int32)
        masked[:3] = (mmax, 0, 129 if mmax > 129 else 1)
        got, ref = _both(dga, A, SFA, B, SFB, masked)
        assert np.array_equal(got, ref), f"m_max {mmax}: {int((got != ref).sum())} of {got.size} outputs differ"


def test_nan_does_not_cross_tiles(dga, oracle):
    g, mmax, n, k = 48, 128, 2048, 384
    A, SFA, B, SFB = _stack(oracle, g, mmax, n, k, seed=31)
    A = A.copy(); A[0, 3, 200] = 0x7F
    masked = np.resize(ALL_L, g).astype(np.int32); masked[0] = 40
    got, ref = _both(dga, A, SFA, B, SFB, masked)
    nan = (got & 0x7FFF) > 0x7F80
    assert np.array_equal(nan, (ref & 0x7FFF) > 0x7F80) and nan.sum() == n
    assert np.array_equal(got[~nan], ref[~nan])


def test_config4_full_size_random_masks(dga):
    """BASELINE configs[3] at full size: 256 experts x (M <= 128, K = 7168, N = 2048), random masks, both kernels byte for byte."""
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    g, mmax, n, k = 256, 128, 2048, 7168
    gen = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(0, 120, (g, mmax, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (g, n, k), dtype=torch.uint8, device="cuda", generator=gen)
    sfa = torch.rand((g, mmax, k // 128), device="cuda") + 0.5
    sfb = torch.rand((g, n // 128, k // 128), device="cuda") + 0.5
    masked = torch.randint(0, mmax + 1, (g,), generator=torch.Generator().manual_seed(3)).to(torch.int32).cuda()
    outs = []
    for grouped_kernel in (True, False):
        out = torch.full((g, mmax, n), -7.0, dtype=torch.bfloat16, device="cuda")
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, expected_m=mmax, policy="bf16_exact", sync=True,
                                                  tiling_=_tiling(dga, mmax, n, k, g, grouped_kernel))
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))


def test_fuzzed_layouts_are_bit_identical_to_the_one_tile_build(dga, oracle):
    """Seeded fuzz over the layout's parameters -- experts 1..300 (rasters below, at and above the CU count), m_max 65..256, N with ragged
    last tiles, K of 2..12 k blocks with and without a tail, masks of every kind (uniform, mostly empty, mostly full, all zero) -- both
    kernels byte for byte, rows at or beyond masked_m untouched."""
    rng = np.random.default_rng(2024)
    init = _bits(torch.tensor([-7.0], dtype=torch.bfloat16))[0]
    for case in range(14):
        g = int(rng.choice([1, 2, 7, 31, 33, 64, 100, 300]))
        mmax = int(rng.choice([65, 96, 128, 128, 128, 200, 256]))
        n = int(rng.choice([256, 384, 512, 1024, 1280, 2048]))
        k = int(rng.choice([256, 384, 512, 640, 1024, 1536])) + int(rng.choice([0, 0, 16, 64]))
        kind = case % 5
        if kind == 0:
            masked = rng.integers(0, mmax + 1, size=g)
        elif kind == 1:
            masked = np.where(rng.random(g) < 0.7, 0, rng.integers(0, mmax + 1, size=g))
        elif kind == 2:
            masked = np.where(rng.random(g) < 0.7, mmax, rng.integers(0, mmax + 1, size=g))
        elif kind == 3:
            masked = rng.integers(0, 20, size=g)
        else:
            masked = np.zeros(g, dtype=np.int64) if case % 2 else rng.integers(60, 70, size=g)
        masked = masked.astype(np.int32)
        A, SFA, B, SFB = _stack(oracle, min(g, 6), mmax, n, k, seed=1000 + case)
        reps = -(-g // A.shape[0])
        A, SFA, B, SFB = (np.concatenate([x] * reps)[:g] for x in (A, SFA, B, SFB))
        got, ref = _both(dga, A, SFA, B, SFB, masked)
        what = f"case {case}: g {g} m_max {mmax} n {n} k {k} mask kind {kind}"
        assert np.array_equal(got, ref), f"{what}: {int((got != ref).sum())} of {got.size} outputs differ"
        for i in range(g):
            assert (got[i, int(masked[i]):] == init).all(), f"{what}: rows >= masked_m were written (expert {i})"


@pytest.mark.parametrize("k", [256, 640 + 16, 7168])
def test_indexed_rows_equal_the_packed_one_tile_build(dga, oracle, k):
    """The indexed form (rows, their scales and their result rows found through the slot table, where they lie in one flat payload buffer:
    dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed) on this kernel against the PACKED call of the one-tile build on the same rows
    copied into the [G, m_max, K] layout: bit for bit; rows nobody owns stay untouched.  40 experts (320 tiles: workgroups walk two),
    every m-tile count, scrambled row order."""
    g, m_max, n = 40, 128, 2048
    kb = -(-k // 128)
    row_bytes = -(-(k + 4 * kb + 4) // 16) * 16
    rng = np.random.default_rng(k)
    masked = np.resize(ALL_L, g).astype(np.int32)
    rows = int(masked.sum()) + 7
    payload = rng.integers(0, 120, size=(rows, row_bytes), dtype=np.uint8)
    sf = rng.uniform(0.5, 1.5, size=(rows, kb)).astype(np.float32)
    sf_off = -(-k // 4) * 4
    payload[:, sf_off:sf_off + 4 * kb] = sf.view(np.uint8)
    b = rng.integers(0, 120, size=(g, n, k), dtype=np.uint8)
    sfb = rng.uniform(0.5, 1.5, size=(g, n // 128, kb)).astype(np.float32)
    perm = rng.permutation(rows)
    row_index = np.full((g, m_max), -1, np.int64)
    at = 0
    for i in range(g):
        row_index[i, :masked[i]] = perm[at:at + masked[i]]
        at += masked[i]
    a_packed = np.zeros((g, m_max, k), np.uint8); sfa_packed = np.ones((g, m_max, kb), np.float32)
    for i in range(g):
        a_packed[i, :masked[i]] = payload[row_index[i, :masked[i]], :k]
        sfa_packed[i, :masked[i]] = sf[row_index[i, :masked[i]]]
    tb, tsfb, tm = _dev(b), _dev(sfb), _dev(masked)
    out_packed = torch.zeros((g, m_max, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((_dev(a_packed), _dev(sfa_packed)), (tb, tsfb), out_packed, tm, m_max, policy="bf16_exact",
                                              tiling_=_tiling(dga, m_max, n, k, g, False), sync=True)
    tp = _dev(payload)
    out_rows = torch.full((rows, n), -5.0, dtype=torch.bfloat16, device="cuda")
    t = _tiling(dga, m_max, n, k, g, True)
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(tp, tp, sf_off, row_bytes // 4, (tb, tsfb), out_rows, _dev(row_index.reshape(-1)),
                                                      tm, m_max, m_max, policy="bf16_exact", tiling_=t, sync=True)
    got, want = _bits(out_rows), _bits(out_packed)
    owned = np.zeros(rows, bool)
    for i in range(g):
        idx = row_index[i, :masked[i]]
        owned[idx] = True
        assert np.array_equal(got[idx], want[i, :masked[i]]), f"group {i} (masked_m {masked[i]})"
    assert (got[~owned] == _bits(torch.tensor([-5.0], dtype=torch.bfloat16))[0]).all()
