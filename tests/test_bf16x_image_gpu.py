"""The image build of the bf16-exact policy (csrc/gemm_fp8_bf16x_image_kernel.hpp: 128 x 256 tile, one wave per SIMD, both
operands converted once per workgroup into a bf16 LDS image) against the in-register build of the same tile and the CPU oracle.

The two builds run the same arithmetic -- four chained v_mfma_f32_16x16x32_bf16 per 128-wide scale block with the same k
placement, then the same fp32 promotion (the counterpart of the reference's device K-loop,
/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:123-369, held to the fp32 golden of
framework/tests/test.py:19-64) -- so the bar between them is BIT IDENTITY; against the oracle it is the policy's bar
(tests/test_bf16_exact_gpu.py).  Two image builds: 8 waves (two per SIMD, 64 x 64 wave tiles: the default of the 128 x 256 tile)
and 4 waves (one per SIMD, 64 x 128).  A third build shares only the A-matrix tile through the image
(gemm_fp8_bf16x_aimage_kernel.hpp, 8 waves; B-matrix fragments converted in registers): `waves` = 1 below.  The builds run only when
a tiling NAMES them through a stage count no tile build has (4 = A image, 5 / 6 = both operands on 8 / 4 waves); every other tiling
of the policy -- 2 x 2 waves and two stages included -- keeps the in-register build.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev, EPS, EPS_ARBITRARY

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent


IMAGES = [8, 4, 1]   # waves of the image build under test (1 = the A-image build)


def _tiling(dga, m, n, k, image, splitk=1, raster=4, groups=1):
    """image: 0 = in-register build, 4 / 8 = image build with that many waves, 1 = A-image build"""
    t = dga.tiling(m, n, k, groups=groups) if groups > 1 else dga.tiling(m, n, k)
    t.m1, t.n1, t.splitkFactor, t.kernelSerial = 128, 256, splitk, (4 if splitk > 1 else 0)
    t.dispatchPolicyTag = 7
    t.stages, t.build = 3, {0: 0, 1: 4, 8: 5, 4: 6}[image]   # (include/dga_hip.h DGA_BUILD_BX_AIMAGE / _IMAGE8 / _IMAGE4)
    t.swizzleOffset = raster
    t.wavesM, t.wavesN = 2, 4
    return t


def _run(dga, a, sfa, b, sfb, t):
    out = torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, policy="bf16_exact", sync=True, tiling_=t)
    return _bits(out)


@pytest.mark.parametrize("m,n,k", [
    (128, 256, 128),      # one tile, one k block
    (128, 128, 128),      # BASELINE configs[0]'s shape: half a tile wide
    (256, 512, 1024),     # 2 x 2 full tiles
    (333, 520, 1168),     # rows, columns and the k block all cut (k % 128 = 16)
    (1, 8, 16),           # one row, eight columns, one 16-byte chunk of K
    (129, 257, 2048),     # one row / one column into the second tile
    (100, 700, 1296),
    (512, 1024, 7168),    # configs[2]'s K
    (64, 256, 8192 + 64),
])
@pytest.mark.parametrize("waves", IMAGES)
def test_image_build_is_bit_identical_to_the_in_register_build(dga, oracle, m, n, k, waves):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m * 13 + n * 5 + k)
    got_img = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, waves))
    got_reg = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, 0))
    assert np.array_equal(got_img, got_reg), f"{int((got_img != got_reg).sum())} of {got_img.size} outputs differ"
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    _assert_bar(oracle, got_img, want, a, sfa, b, sfb, eps=EPS if k >= 128 else EPS_ARBITRARY)


@pytest.mark.parametrize("waves", IMAGES)
@pytest.mark.parametrize("raster", [1, 2, 8])
def test_rasters(dga, oracle, raster, waves):
    m, n, k = 640, 1280, 512   # 5 x 5 tiles: the grouped raster's last band is short
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=raster)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, waves, raster=raster))
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, 0, raster=raster))
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("waves", IMAGES)
@pytest.mark.parametrize("splitk", [2, 3, 5])
def test_split_k(dga, oracle, splitk, waves):
    m, n, k = 130, 300, 4096 + 128   # 33 k blocks: the last split is short
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=splitk)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, waves, splitk=splitk))
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, 0, splitk=splitk))
    assert np.array_equal(got, ref)
    _assert_bar(oracle, got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)


@pytest.mark.parametrize("waves", IMAGES)
def test_arbitrary_bytes_wild_scales_and_nan(dga, oracle, waves):
    m, n, k = 192, 256, 640
    rng = np.random.default_rng(5)
    a = oracle.random_fp8_bytes((m, k), seed=1)
    b = oracle.random_fp8_bytes((n, k), seed=2)
    a[3, 17] = 0x7F; b[100, 200] = 0xFF
    sfa = np.exp2(rng.uniform(-30, 4, size=(m, 5))).astype(np.float32)
    sfb = np.exp2(rng.uniform(-30, 4, size=(2, 5))).astype(np.float32)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, waves))
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, 0))
    nan = (got & 0x7FFF) > 0x7F80
    assert np.array_equal(nan, (ref & 0x7FFF) > 0x7F80)
    assert np.array_equal(got[~nan], ref[~nan])
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    _assert_bar(oracle, got, want, a, sfa, b, sfb, eps=EPS_ARBITRARY)


@pytest.mark.parametrize("waves", IMAGES)
def test_every_e4m3_code_converts_exactly(dga, oracle, waves):
    codes = np.array([c for c in range(256) if (c & 0x7F) != 0x7F], np.uint8)
    m = n = codes.size
    k = 128
    a = np.zeros((m, k), np.uint8)
    a[np.arange(m), np.arange(m) % k] = codes
    b = np.repeat(codes[:, None], k, axis=1)
    sfa = np.ones((m, 1), np.float32); sfb = np.ones(((n + 127) // 128, 1), np.float32)
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, waves))
    assert np.array_equal(got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4))


def test_grouped_masked(dga, oracle):
    """Masked grouped layout on the image build: rows at or beyond masked_m[g] stay untouched, empty experts are skipped."""
    g, mmax, n, k = 6, 128, 512, 1024
    parts = [oracle.make_inputs(mmax, n, k, seed=70 + i) for i in range(g)]
    A, SFA, B, SFB = (np.stack([p[j] for p in parts]) for j in range(4))
    masked = np.array([128, 0, 1, 77, 127, 64], np.int32)
    outs = []
    for image in (8, 0, 4, 1):
        out = torch.full((g, mmax, n), -7.0, dtype=torch.bfloat16, device="cuda")
        t = _tiling(dga, mmax, n, k, image, groups=g)
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((_dev(A), _dev(SFA)), (_dev(B), _dev(SFB)), out, _dev(masked),
                                                  expected_m=64, policy="bf16_exact", sync=True, tiling_=t)
        outs.append(_bits(out))
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[2], outs[1]) and np.array_equal(outs[3], outs[1])
    init = np.full((g, mmax, n), _bits(torch.tensor([-7.0], dtype=torch.bfloat16))[0], np.uint16)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(A, SFA, B, SFB, init, masked, threads=8)
    for i in range(g):
        mm = int(masked[i])
        assert np.array_equal(outs[0][i, mm:], init[i, mm:]), "rows >= masked_m were written"
        if mm:
            _assert_bar(oracle, outs[0][i, :mm], want[i, :mm], A[i, :mm], SFA[i, :mm], B[i], SFB[i])


@pytest.mark.parametrize("shape", ["dense_4096", "dsv3_prefill"])
def test_baseline_configs_bit_identical_at_full_size(dga, shape):
    """BASELINE configs[1] and configs[2] on bench.py's inputs: every output of the image build equals the in-register build's."""
    sys.path.insert(0, str(ROOT))
    import bench
    m, n, k = bench.WORKLOADS[shape]
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    outs = []
    for image in (8, 0, 4, 1):
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", sync=True, tiling_=_tiling(dga, m, n, k, image))
        outs.append(out)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[2].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[3].view(torch.int16), outs[1].view(torch.int16))
