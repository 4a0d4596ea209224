"""CPU: the oracle against the committed golden vectors and against the reference's golden formula."""
import json
from pathlib import Path

import numpy as np
import pytest

G = Path(__file__).parent / "golden"


def test_e4m3fn_table_matches_fixture(oracle):
    fx = json.loads((G / "e4m3fn_table.json").read_text())
    want = np.array([int(h, 16) for h in fx["f32_bits_hex"]], np.uint32).view(np.float32)
    for tab in (oracle.e4m3fn_table(), oracle.np_e4m3fn_table()):
        assert np.array_equal(tab.view(np.uint32) & 0x7FFFFFFF | (tab.view(np.uint32) & 0x80000000),
                              want.view(np.uint32)) or np.array_equal(tab, want, equal_nan=True)
    # the cases SURVEY.md section 7 calls out: 0x7F / 0xFF NaN, 0x80 = -0, 0x7E = 448, 0x01 = 2^-9
    t = oracle.e4m3fn_table()
    assert np.isnan(t[0x7F]) and np.isnan(t[0xFF])
    assert t[0x80] == 0 and np.signbit(t[0x80])
    assert t[0x7E] == 448.0 and t[0xFE] == -448.0 and t[0x01] == 2.0 ** -9 and t[0x08] == 2.0 ** -6


def test_encode_roundtrip_and_rne(oracle):
    L = oracle.lib()
    t = oracle.e4m3fn_table()
    for v in range(256):
        if np.isnan(t[v]):
            continue
        assert L.dga_oracle_f32_to_e4m3fn(float(t[v])) == (v if t[v] != 0 else (v & 0x80))
    assert L.dga_oracle_f32_to_e4m3fn(17.0) == L.dga_oracle_f32_to_e4m3fn(16.0)  # tie 17 (16|18) -> even 16
    assert t[L.dga_oracle_f32_to_e4m3fn(19.0)] == 20.0                           # tie 19 (18|20) -> even 20
    assert t[L.dga_oracle_f32_to_e4m3fn(1e6)] == 448.0                            # satfinite


def test_bf16_rounding(oracle):
    x = np.array([1.0, 1.00390625, 1.01171875, -3.14159, 65504.0, 1e-40, np.inf, np.nan], np.float32)
    got = oracle.f32_to_bf16_bits(x)
    c = np.array([oracle.lib().dga_oracle_f32_to_bf16(float(v)) for v in x], np.uint16)
    assert np.array_equal(got, c)
    assert got[0] == 0x3F80 and got[1] == 0x3F80 and got[2] == 0x3F82  # ties to even, both directions
    assert (got[-1] & 0x7FFF) > 0x7F80


def test_config1_golden_is_the_reference_formula(oracle):
    """BASELINE config 1 (128^3, unit scales): oracle == np.matmul(f32,f32) of the decoded values
    (/root/reference/deep_gemm_ascend/framework/tests/test.py:37), bit for bit -- K = 128 is one
    scale block and every fp8 x fp8 product sum here is exact in fp32."""
    fx = np.load(G / "c1_unit_128.npz")
    out, acc = oracle.gemm_fp8_fp8_bf16_nt(fx["a"], fx["sfa"], fx["b"], fx["sfb"], want_f32=True)
    assert np.array_equal(acc, fx["golden_f32"])
    assert np.array_equal(out, fx["expected_bf16"])
    # the same through the oracle's restatement of the reference NN matmul
    tab = oracle.e4m3fn_table()
    nn = oracle.matmul_f32_nn(tab[fx["a"]], tab[fx["b"]].T.copy())
    assert np.array_equal(nn, fx["golden_f32"])


def test_scaled_fixture_and_numpy_cross_check(oracle):
    fx = np.load(G / "scaled_64x256x400.npz")
    out, acc = oracle.gemm_fp8_fp8_bf16_nt(fx["a"], fx["sfa"], fx["b"], fx["sfb"], want_f32=True, threads=2)
    assert np.array_equal(out, fx["expected_bf16"])
    assert np.array_equal(acc, fx["acc_f32"])
    # independent numpy restatement (BLAS summation order differs: compare in bf16 ulps)
    acc_np = oracle.np_gemm_fp8_fp8_bf16_nt(fx["a"], fx["sfa"], fx["b"], fx["sfb"])
    assert oracle.bf16_ulp_diff(out, oracle.f32_to_bf16_bits(acc_np)).max() <= 1
    # fp64 tie-break reference
    f64 = oracle.gemm_fp8_fp8_f64_nt(fx["a"], fx["sfa"], fx["b"], fx["sfb"])
    assert oracle.bf16_ulp_diff(out, oracle.f32_to_bf16_bits(f64.astype(np.float32))).max() <= 1


def test_grouped_fixture(oracle):
    fx = np.load(G / "grouped_g4_m16.npz")
    out = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(fx["a"], fx["sfa"], fx["b"], fx["sfb"], fx["init"], fx["masked_m"])
    assert np.array_equal(out, fx["expected_bf16"])
    for g, mm in enumerate(fx["masked_m"]):
        assert (out[g, mm:] == 0x7FC1).all()                    # rows >= masked_m untouched
        dense = oracle.gemm_fp8_fp8_bf16_nt(fx["a"][g, :mm], fx["sfa"][g, :mm], fx["b"][g], fx["sfb"][g])
        assert np.array_equal(out[g, :mm], dense)


def test_edge_shapes(oracle):
    a, sfa, b, sfb = oracle.make_inputs(0, 128, 256, seed=1)
    assert oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb).shape == (0, 128)
    a, sfa, b, sfb = oracle.make_inputs(3, 5, 0, seed=1)
    assert (oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb) == 0).all()  # empty sum
    a, sfa, b, sfb = oracle.make_inputs(1, 1, 129, seed=1)              # K tail of one element
    out = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb)
    f64 = oracle.gemm_fp8_fp8_f64_nt(a, sfa, b, sfb)
    assert oracle.bf16_ulp_diff(out, oracle.f32_to_bf16_bits(f64.astype(np.float32))).max() <= 1


def test_nan_semantics(oracle):
    a, sfa, b, sfb = oracle.make_inputs(4, 130, 128, seed=2)
    a[1, 5] = 0x7F
    out = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb)
    nan = (out & 0x7FFF) > 0x7F80
    assert nan[1].all() and nan.sum() == 130


def test_verifier_restatement(oracle):
    """scripts/verify.py semantics incl. the two cases where the reference itself breaks under numpy 2
    (size mismatch => fail; empty => pass), SURVEY.md section 4."""
    g = np.linspace(1, 2, 1000, dtype=np.float32)
    ok, r = oracle.verify_isclose(g + 1e-8, g, rtol=1e-6)
    assert ok and r == 0
    ok, r = oracle.verify_isclose(g * 2, g, rtol=1e-6)
    assert not ok and r == 1.0
    o = g.copy(); o[:1] = 5  # 1e-3 mismatch fraction > 1e-4
    assert not oracle.verify_isclose(o, g, rtol=1e-6)[0]
    n = g.copy(); n[3] = np.nan
    assert oracle.verify_isclose(n, n, rtol=1e-6)[0]      # NaN == NaN
    assert not oracle.verify_isclose(g[:10], g[:9], rtol=1e-6)[0]
    assert oracle.verify_isclose(g[:0], g[:0], rtol=1e-6)[0]


def test_parity_metric_behaves(oracle):
    a, sfa, b, sfb = oracle.make_inputs(32, 128, 256, seed=4)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb)
    oracle.assert_parity(want, want, a, sfa, b, sfb)
    bad = want.copy(); bad[3, 7] ^= 0x0040  # flip a high mantissa bit: ~64 ulp
    with pytest.raises(AssertionError):
        oracle.assert_parity(bad, want, a, sfa, b, sfb)


def test_contiguous_layout_is_rowwise_dense(oracle):
    """The contiguous-grouped oracle equals the dense oracle applied per row group; padding rows keep out_init."""
    k, n = 256, 128
    a, sfa, _, _ = oracle.make_inputs(40, 8, k, seed=1)
    bs = [oracle.make_inputs(8, n, k, seed=10 + g)[2:] for g in range(3)]
    b = np.stack([x[0] for x in bs]); sfb = np.stack([x[1] for x in bs])
    idx = np.array([0] * 10 + [-1] * 6 + [2] * 16 + [1] * 8, np.int32)
    init = np.full((40, n), 0x7FC1, np.uint16)
    out = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a, sfa, b, sfb, init, idx)
    assert (out[10:16] == 0x7FC1).all()
    for g, rows in ((0, slice(0, 10)), (2, slice(16, 32)), (1, slice(32, 40))):
        assert (out[rows] == oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b[g], sfb[g])).all()
