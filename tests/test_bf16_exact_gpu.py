"""The bf16-exact dispatch policy (dispatchPolicyTag 7) against the CPU oracle and the strict kernel.

BASELINE.json's north_star asks for "within 2 ULP bf16" of the reference CPU path (fp32 accumulate,
/root/reference/deep_gemm_ascend/framework/tests/test.py:19-64).  The policy up-converts the e4m3 bytes to bf16 in
registers (exact) and sums every 128-wide scale block on v_mfma_f32_16x16x32_bf16: exact products, fp32-class sums in
the instruction's own order.  Its bar, asserted here:

  * PLAIN bound |got - want| <= 2 ulp_bf16(want) on at least 99.999 % of the elements (on small samples: all but 2);
  * the remainder -- sums that cancel to less than ~2^-21 of their terms (measured on configs[1]: scripts/bf16_exact_outliers.py), where the oracle's own fp32 rounding is as
    large as the difference -- within 2 ulp + 2^-22 * S (S = sum of the magnitudes of the scaled products; 2^-19 for
    arbitrary bit patterns), i.e. 128 times tighter than the fast path's 2^-15;
  * NaN positions identical.
"""
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent
EPS = 2.0 ** -22
EPS_ARBITRARY = 2.0 ** -19
FRAC = 1e-5


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _assert_bar(oracle, got, want, a, sfa, b, sfb, eps=EPS):
    rep = oracle.parity_report(got, want, a, sfa, b, sfb)
    assert rep["nan_positions_equal"], "NaN positions differ"
    size = int(np.asarray(got).size)
    beyond = int(round(rep["frac_gt_max_ulp"] * size))
    assert beyond <= max(FRAC * size, 2), f"{beyond} of {size} elements beyond 2 ulp: {rep}"
    assert rep["worst_excess_over_S"] <= eps, f"excess {rep['worst_excess_over_S']:.3e} * S > {eps:.3e}: {rep}"
    return rep


def _run(dga, a, sfa, b, sfb, tiling_=None):
    out = torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, policy="bf16_exact", sync=True,
                             tiling_=tiling_)
    return _bits(out)


def test_config1_unit_scales_is_the_reference_golden(dga, oracle):
    """BASELINE configs[0]: 128^3, unit scales = the reference golden np.matmul(f32, f32) (test.py:37), bf16-rounded."""
    fx = np.load(ROOT / "tests" / "golden" / "c1_unit_128.npz")
    a, b = fx["a"], fx["b"]
    sfa = np.ones((128, 1), np.float32); sfb = np.ones((1, 1), np.float32)
    got = _run(dga, a, sfa, b, sfb)
    d = oracle.bf16_ulp_diff(got, fx["expected_bf16"])
    assert int(d.max()) <= 1, f"max ulp {int(d.max())} against the committed golden"
    assert int(oracle.bf16_ulp_diff(got, oracle.f32_to_bf16_bits(fx["golden_f32"])).max()) <= 1


@pytest.mark.parametrize("m,n,k", [
    (128, 128, 128), (64, 256, 384), (300, 200, 256), (1, 128, 128), (7, 136, 1024), (129, 257, 144),
    (512, 384, 7168), (33, 4096, 512), (256, 512, 2048), (31, 127, 2048), (100, 700, 1296),
    (40, 130, 100), (17, 33, 7), (64, 128, 0),     # K % 16 != 0 (padding pass / element-wise kernel), K = 0
])
def test_dense_against_the_oracle(dga, oracle, m, n, k):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m * 31 + n * 7 + k)
    got = _run(dga, a, sfa, b, sfb)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    _assert_bar(oracle, got, want, a, sfa, b, sfb, eps=EPS if k >= 128 else EPS_ARBITRARY)


@pytest.mark.parametrize("m1,n1", [(128, 256), (128, 128), (64, 256), (64, 128), (32, 128)])
def test_every_tile_of_the_menu(dga, oracle, m1, n1):
    """Each build of the policy's menu on one ragged problem (tile rows / columns / k tail all cut)."""
    m, n, k = 333, 520, 1168
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m1 * 3 + n1)
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.splitkFactor, t.kernelSerial = m1, n1, 1, 0
    got = _run(dga, a, sfa, b, sfb, tiling_=t)
    _assert_bar(oracle, got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)


def test_split_k(dga, oracle):
    m, n, k = 48, 640, 8192
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=9)
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.splitkFactor, t.kernelSerial = 64, 128, 4, 4
    got = _run(dga, a, sfa, b, sfb, tiling_=t)
    _assert_bar(oracle, got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)


def test_arbitrary_bytes_wild_scales_and_nan(dga, oracle):
    """All e4m3fn encodings incl. subnormals, -0, NaN; scales over many binades."""
    m, n, k = 192, 256, 640
    rng = np.random.default_rng(5)
    a = oracle.random_fp8_bytes((m, k), seed=1)
    b = oracle.random_fp8_bytes((n, k), seed=2)
    a[3, 17] = 0x7F; b[100, 200] = 0xFF
    sfa = np.exp2(rng.uniform(-30, 4, size=(m, 5))).astype(np.float32)
    sfb = np.exp2(rng.uniform(-30, 4, size=(2, 5))).astype(np.float32)
    got = _run(dga, a, sfa, b, sfb)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    assert ((want & 0x7FFF) > 0x7F80).sum() == n + m - 1
    _assert_bar(oracle, got, want, a, sfa, b, sfb, eps=EPS_ARBITRARY)


def test_every_e4m3_code_converts_exactly(dga, oracle):
    """One-hot rows: out[m, n] = value(code_a[m]) * value(code_b[n]) -- every pair of the 254 finite codes, with unit
    scales, has to come out as the bf16 rounding of the exact product (the in-register conversion is exact)."""
    codes = np.array([c for c in range(256) if (c & 0x7F) != 0x7F], np.uint8)
    m = n = codes.size
    k = 128
    a = np.zeros((m, k), np.uint8); b = np.zeros((n, k), np.uint8)
    a[np.arange(m), np.arange(m) % k] = codes
    b[:, :] = 0
    # every B row holds its code at EVERY k, so that whichever k the A row is hot at meets it
    b[:, :] = codes[:, None]
    sfa = np.ones((m, 1), np.float32); sfb = np.ones(((n + 127) // 128, 1), np.float32)
    got = _run(dga, a, sfa, b, sfb)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("shape", ["c2", "c3"])
def test_baseline_configs_on_their_own_recipe(dga, oracle, shape):
    """BASELINE configs[1] (4096^3) and configs[2] (M=4096, K=7168, N=2048) on bench.py's amax-quantised inputs: 256
    sampled rows against the CPU oracle, and ALL outputs against the strict kernel (pinned bit for bit to the oracle in
    tests/test_strict_gpu.py) with S computed on the device."""
    sys.path.insert(0, str(ROOT))
    import bench
    m, n, k = bench.WORKLOADS["dense_4096" if shape == "c2" else "dsv3_prefill"]
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", sync=True)
    rows = np.arange(5, m, 16)[:256]
    an, san, bn, sbn = a.cpu().numpy(), sfa.cpu().numpy(), b.cpu().numpy(), sfb.cpu().numpy()
    want = oracle.gemm_fp8_fp8_bf16_nt(an[rows], san[rows], bn, sbn, threads=16)
    _assert_bar(oracle, _bits(out[torch.from_numpy(rows).cuda()]), want, an[rows], san[rows], bn, sbn)
    rep = bench.parity_vs_strict(dga, a, sfa, b, sfb, out)
    assert rep["frac_gt_2ulp"] <= FRAC, rep
    assert rep["worst_excess_over_S"] <= EPS * 1.01, rep   # 1.01: S itself is bf16-rounded (2^-9)


@pytest.mark.parametrize("tile", [None, (32, 128), (64, 256), (128, 256)])
def test_grouped_masked_c4_experts(dga, oracle, tile):
    """BASELINE configs[3] shape per expert (M<=128, K=7168, N=2048), 8 experts with ragged masks; masked rows stay untouched.
    tile None: what the policy's selector names whatever the expected_m hint says -- the layout's own kernel on the 128 x 256 tile
    (DGA_BUILD_BX_GROUPED; until round 5 the tile height followed the hint because the loop multiplied every row of its tile).  The
    other cases: the tile builds a caller can still name for this layout (a tile lower than the rows present costs time, never rows)."""
    g, mmax, n, k = 8, 128, 2048, 7168
    for hint in (16, 64, 128):
        t = dga.tiling(mmax, n, k, groups=g, expected_m=hint, policy="bf16_exact")
        assert (t.m1, t.n1, t.build, t.kernelSerial, t.splitkFactor) == (128, 256, 9, 0, 1), t.as_dict()
    if tile is not None:
        t.m1, t.n1, t.build = tile[0], tile[1], (8 if tile == (128, 256) else 0)
        t.blockDim = g * -(-mmax // t.m1) * -(-n // t.n1)
    parts = [oracle.make_inputs(mmax, n, k, seed=40 + i) for i in range(g)]
    A, SFA, B, SFB = (np.stack([p[j] for p in parts]) for j in range(4))
    masked = np.array([128, 0, 1, 77, 128, 16, 127, 64], np.int32)
    out = torch.full((g, mmax, n), -7.0, dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((_dev(A), _dev(SFA)), (_dev(B), _dev(SFB)), out, _dev(masked),
                                              expected_m=64, policy="bf16_exact", sync=True, tiling_=None if tile is None else t)
    got = _bits(out)
    init = np.full((g, mmax, n), _bits(torch.tensor([-7.0], dtype=torch.bfloat16))[0], np.uint16)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(A, SFA, B, SFB, init, masked, threads=8)
    for i in range(g):
        mm = int(masked[i])
        assert np.array_equal(got[i, mm:], init[i, mm:]), "rows >= masked_m were written"
        if mm:
            _assert_bar(oracle, got[i, :mm], want[i, :mm], A[i, :mm], SFA[i, :mm], B[i], SFB[i])


def test_contiguous_layout(dga, oracle):
    groups, n, k = 3, 384, 512
    idx = np.concatenate([np.full(128, 0), np.full(70, 2), np.full(58, -1), np.full(128, 1)]).astype(np.int32)
    msum = idx.size
    a, sfa, _, _ = oracle.make_inputs(msum, n, k, seed=1)
    bs = [oracle.make_inputs(8, n, k, seed=10 + g)[2:] for g in range(groups)]
    b = np.stack([x[0] for x in bs]); sfb = np.stack([x[1] for x in bs])
    out = torch.full((msum, n), -3.0, dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, _dev(idx),
                                                  policy="bf16_exact", sync=True)
    init = np.full((msum, n), _bits(torch.tensor([-3.0], dtype=torch.bfloat16))[0], np.uint16)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a, sfa, b, sfb, init, idx, threads=8)
    got = _bits(out)
    assert np.array_equal(got[idx < 0], init[idx < 0])
    for g in range(groups):
        sel = idx == g
        _assert_bar(oracle, got[sel], want[sel], a[sel], sfa[sel], b[g], sfb[g])


def test_policy_argument_is_checked(dga):
    t = torch.zeros((128, 128), dtype=torch.uint8, device="cuda")
    sf = torch.ones((128, 1), dtype=torch.float32, device="cuda")
    sfb = torch.ones((1, 1), dtype=torch.float32, device="cuda")
    out = torch.zeros((128, 128), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(dga.DGAError):
        dga.gemm_fp8_fp8_bf16_nt((t, sf), (t, sfb), out, policy="exactish")
    with pytest.raises(dga.DGAError):
        dga.gemm_fp8_fp8_bf16_nt((t, sf), (t, sfb), out, strict=True, policy="bf16_exact")


def test_a_default_call_runs_the_tag_7_row_of_the_cache(dga, oracle, tmp_path):
    """A cache row written under this policy (dispatchPolicyTag 7: harness/sweep.py --arith bf16_exact) is the tiling of a call that names
    neither policy nor tiling -- the reference's op consults its cache first (select_kernel.cpp:371-378) -- and the result holds the bar."""
    m, n, k = 300, 520, 1024
    path = tmp_path / "bx.csv"
    path.write_text("m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,splitkFactor,stages,swizzleOffset,wavesM,wavesN,"
                    "dispatchPolicyTag,groups,contiguous,build\n"
                    f"{m},{n},{k},64,256,128,4,0,0,0,30,2,3,1,0,0,7,1,0,0\n")
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=21)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    try:
        dga.tiling_cache_open(str(path))
        dga.api._PLANS.clear()
        t = dga.api._planned(0, m, n, k, 1, 0, False, False, None)
        assert (t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.dispatchPolicyTag) == (64, 256, 2, 4, 7)
        out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, sync=True)
        _assert_bar(oracle, _bits(out), want, a, sfa, b, sfb)
    finally:
        dga.tiling_cache_open(None)
        dga.tiling_cache_clear()
        dga.api._PLANS.clear()
