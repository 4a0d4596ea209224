"""The persistent form of the bf16-exact policy's 128 x 256 build (csrc/gemm_fp8_bf16x_persistent_kernel.hpp: one workgroup per CU walks
the raster, the LDS ring runs across tile boundaries) against the one-tile build of the same tile: the same arithmetic in the same order,
so the bar is BIT IDENTITY; against the oracle it is the policy's bar (tests/test_bf16_exact_gpu.py).  Counterpart in the reference:
its device loop walks the tiles of a core's section with double-buffered L1 across them
(/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:160-198).  tiling.build = 7 (DGA_BUILD_BX_PERSISTENT) names the persistent form, 8 (_ONE_TILE)
the one-tile build (0 = the dispatcher's rule: persistent on every raster of more than one round)."""
import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev, EPS, EPS_ARBITRARY

pytestmark = pytest.mark.gpu


def _tiling(dga, m, n, k, persistent, groups=1, raster=4):
    t = dga.tiling(m, n, k, groups=groups, policy="bf16_exact") if groups > 1 else dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.dispatchPolicyTag = 128, 256, 1, 0, 7
    t.stages, t.build, t.wavesM, t.wavesN, t.swizzleOffset = 3, (7 if persistent else 8), 2, 4, raster
    return t


def _run(dga, a, sfa, b, sfb, t):
    out = torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, policy="bf16_exact", sync=True, tiling_=t)
    return _bits(out)


@pytest.mark.parametrize("m,n,k", [
    (128, 256, 256),            # one tile, two k blocks
    (2304, 4096, 384),          # 288 tiles on 256 workgroups: 32 of them walk two tiles
    (2400, 4200, 400),          # the same with every edge cut (rows, columns, K % 128 = 16)
    (129, 257, 2048),           # one row / one column into the second tile: three tiles are almost empty
    (4096, 4096, 256),          # 512 tiles, two per workgroup
    (4096, 7168, 256),          # 896 tiles = 3.5 rounds: the uneven rasters the dispatcher hands to this kernel since round 5
    (1000, 9000, 384 + 16),     # 8 x 36 = 288 tiles, every edge cut, uneven
    (100, 70000, 256 + 16),     # a short tile row: the second half of the waves has no rows in any tile
    (64, 256, 128 * 9),
])
@pytest.mark.parametrize("raster", [1, 4])
def test_persistent_is_bit_identical_to_the_one_tile_build(dga, oracle, m, n, k, raster):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m * 3 + n + 7 * k)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, True, raster=raster))
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, False, raster=raster))
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ"
    if m * n * k <= 2400 * 4200 * 400:
        want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
        _assert_bar(oracle, got, want, a, sfa, b, sfb, eps=EPS)


def test_one_k_block_keeps_the_one_tile_build(dga, oracle):
    m, n, k = 300, 600, 128
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=1)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, True))
    assert np.array_equal(got, _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, False)))


def test_nan_and_wild_scales_cross_tile_boundaries_cleanly(dga, oracle):
    """A NaN in one tile must not reach the tiles the same workgroup multiplies afterwards (the promotion ring is cleared)."""
    m, n, k = 2304, 4096, 384
    rng = np.random.default_rng(3)
    a = oracle.random_fp8_bytes((m, k), seed=1); b = oracle.random_fp8_bytes((n, k), seed=2)
    a[5, 300] = 0x7F; b[4000, 10] = 0xFF
    sfa = np.exp2(rng.uniform(-20, 4, size=(m, 3))).astype(np.float32)
    sfb = np.exp2(rng.uniform(-20, 4, size=(n // 128, 3))).astype(np.float32)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, True))
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, False))
    nan = (got & 0x7FFF) > 0x7F80
    assert np.array_equal(nan, (ref & 0x7FFF) > 0x7F80) and nan.sum() == n + m - 1
    assert np.array_equal(got[~nan], ref[~nan])


def test_grouped_masked(dga, oracle):
    """Masked grouped layout: 40 experts x 8 n-tiles = 320 tiles on 256 workgroups, ragged masks incl. empty experts and experts whose
    second half of the tile has no rows; rows at or beyond masked_m stay untouched."""
    g, mmax, n, k = 40, 128, 2048, 512
    parts = [oracle.make_inputs(mmax, n, k, seed=90 + i) for i in range(4)]
    A, SFA, B, SFB = (np.stack([parts[i % 4][j] for i in range(g)]) for j in range(4))
    masked = np.array([128, 0, 1, 77, 127, 64, 65, 16] * 5, np.int32)
    outs = []
    for persistent in (True, False):
        out = torch.full((g, mmax, n), -7.0, dtype=torch.bfloat16, device="cuda")
        t = _tiling(dga, mmax, n, k, persistent, groups=g)
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((_dev(A), _dev(SFA)), (_dev(B), _dev(SFB)), out, _dev(masked),
                                                  expected_m=128, policy="bf16_exact", sync=True, tiling_=t)
        outs.append(_bits(out))
    assert np.array_equal(outs[0], outs[1])
    init = np.full((mmax, n), _bits(torch.tensor([-7.0], dtype=torch.bfloat16))[0], np.uint16)
    for i in range(8):
        mm = int(masked[i])
        assert np.array_equal(outs[0][i, mm:], init[mm:]), "rows >= masked_m were written"
        if mm:
            want = oracle.gemm_fp8_fp8_bf16_nt(A[i, :mm], SFA[i, :mm], B[i], SFB[i], threads=8)
            _assert_bar(oracle, outs[0][i, :mm], want, A[i, :mm], SFA[i, :mm], B[i], SFB[i])


@pytest.mark.parametrize("shape", ["dense_4096", "dsv3_prefill"])
def test_baseline_configs_bit_identical_at_full_size(dga, shape):
    import sys
    from pathlib import Path
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    import bench
    m, n, k = bench.WORKLOADS[shape]
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    outs = []
    for persistent in (True, False):
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", sync=True, tiling_=_tiling(dga, m, n, k, persistent))
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
