"""CPU: the product harness states its parity bar once (harness/tolerance.py); the bbit file verifier and the sweep's gate
both reject an output that violates 2 ulp_bf16 + eps * S, and the oracle's test-side statement agrees with it."""
import numpy as np
import pytest

from deepgemm_ascend_amd.harness import files, tolerance


def _case(m=48, n=160, k=384, seed=3):
    rng = np.random.default_rng(seed)
    a, sfa = files.quant_blocks(rng.standard_normal((m, k)).astype(np.float32), 1)
    b, sfb = files.quant_blocks(rng.standard_normal((n, k)).astype(np.float32), 128)
    golden = files.golden_fp8(a, sfa, b, sfb)
    return a, sfa, b, sfb, golden, files.abs_term_sum_fp8(a, sfa, b, sfb)


def test_bar_accepts_the_rounded_golden_and_two_ulp(oracle):
    a, sfa, b, sfb, golden, s = _case()
    want = tolerance.bf16_round(golden)
    assert np.array_equal(want.view(np.uint32) >> 16, oracle.f32_to_bf16_bits(golden))      # the same RNE as the oracle's
    for policy in ("fast", "bf16_exact", "strict"):
        ok, rep = tolerance.check(want, want, s, policy=policy)
        assert ok and rep["max_ulp"] == 0
    two = want + 2 * oracle.bf16_ulp_of(want).astype(np.float32) * np.sign(want)
    assert tolerance.check(two, want, s, policy="strict")[0]                   # strict: the PLAIN bound, no eps, no fraction
    three = want + 3 * oracle.bf16_ulp_of(want).astype(np.float32) * np.sign(want)
    three[1:] = want[1:]
    assert not tolerance.check(three, want, s, policy="strict")[0]
    ok, rep = tolerance.check(two, want, s, policy="fast")
    assert ok and abs(rep["max_ulp"] - 2) < 1e-6 and rep["elements_gt_2ulp"] == 0


@pytest.mark.parametrize("policy,eps", [("fast", 2.0 ** -15), ("bf16_exact", 2.0 ** -22)])
def test_bar_rejects_one_element_past_two_ulp_plus_eps_s(oracle, policy, eps):
    a, sfa, b, sfb, golden, s = _case()
    want = tolerance.bf16_round(golden)
    ulp = oracle.bf16_ulp_of(want)
    inside = want.astype(np.float64).copy()
    inside[5, 7] += 2 * ulp[5, 7] + 0.9 * eps * s[5, 7]
    ok, rep = tolerance.check(inside, want, s, policy=policy)
    assert ok and rep["elements_gt_2ulp"] == 1, rep
    outside = want.astype(np.float64).copy()
    outside[5, 7] += 2 * ulp[5, 7] + 1.5 * eps * s[5, 7]
    ok, rep = tolerance.check(outside, want, s, policy=policy)
    assert not ok and rep["worst_excess_over_S"] > eps, rep
    # the oracle's test-side statement draws the line at the same place
    bits = lambda x: oracle.f32_to_bf16_bits(x.astype(np.float32))
    assert oracle.parity_excess(bits(want), bits(want), a, sfa, b, sfb, eps=eps)[0]


def test_bar_rejects_too_many_elements_past_two_ulp_and_nan_mismatch(oracle):
    a, sfa, b, sfb, golden, s = _case()
    want = tolerance.bf16_round(golden).astype(np.float64)
    ulp = oracle.bf16_ulp_of(want)
    many = want.copy()
    many[:, :8] += 2 * ulp[:, :8] + 2.0 ** -17 * s[:, :8]            # inside eps * S, but 5 % of the elements
    assert not tolerance.check(many, want, s, policy="fast")[0]
    nan = want.copy(); nan[0, 0] = np.nan
    assert not tolerance.check(nan, want, s, policy="fast")[0]


def test_bar_against_a_golden_of_unspecified_summation_order(oracle):
    """The harness's goldens are np.matmul / torch.matmul of the dequantised operands: their own rounding puts 2e-5..6e-5 of the
    outputs of even the bit-exact strict kernel beyond 2 ULP (profiles/r03_golden_order_noise.txt).  golden_order="any" floors
    the bar there (eps 2^-21, frac 2e-4) -- and not further."""
    a, sfa, b, sfb, golden, s = _case(m=128, n=512, k=384)
    want = tolerance.bf16_round(golden).astype(np.float64)
    ulp = oracle.bf16_ulp_of(want.astype(np.float32)).astype(np.float64)
    noisy = want.copy()
    idx = np.unravel_index(np.arange(0, want.size, want.size // 12)[:12], want.shape)    # 12 of 65536 = 1.8e-4 of the elements
    noisy[idx] += 2 * ulp[idx] + 2.0 ** -23 * s[idx]
    for policy in ("strict", "bf16_exact"):
        assert not tolerance.check(noisy, want, s, policy=policy)[0]                      # against an oracle-order reference: rejected
        ok, rep = tolerance.check(noisy, want, s, policy=policy, golden_order="any")
        assert ok and rep["elements_gt_2ulp"] == 12, rep
    worse = want.copy()
    worse[idx] += 2 * ulp[idx] + 2.0 ** -19 * s[idx]                                      # past the floor: still rejected
    assert not tolerance.check(worse, want, s, policy="strict", golden_order="any")[0]
    many = want.copy()
    sel = np.unravel_index(np.arange(0, want.size, 1000), want.shape)                     # 1e-3 of the elements
    many[sel] += 2 * ulp[sel] + 2.0 ** -24 * s[sel]
    assert not tolerance.check(many, want, s, policy="bf16_exact", golden_order="any")[0]
    assert tolerance.check(many, want, s, policy="fast", golden_order="any")[0]          # the fast policy's own bar is wider and stays
    with pytest.raises(ValueError):
        tolerance.check(want, want, s, golden_order="blas")


def test_file_verifier_uses_the_same_bar(tmp_path, monkeypatch, oracle):
    monkeypatch.chdir(tmp_path)
    (a, sfa), (b, sfb), golden = files.gen_golden_data(64, 256, 512, mode="fp8", seed=4)
    s = files.abs_term_sum_fp8(a, sfa, b, sfb)
    want = tolerance.bf16_round(golden)
    to_file = lambda x: oracle.f32_to_bf16_bits(x.astype(np.float32)).tofile("output/output.bin")
    to_file(want)
    assert files.verify_result("output/output.bin", "output/golden.bin", mode="fp8")
    assert files.verify_result("output/output.bin", "output/golden.bin", mode="fp8", policy="strict")
    bad = want.astype(np.float64).copy()
    i = np.unravel_index(np.argmax(s), s.shape)
    bad[i] += 4 * oracle.bf16_ulp_of(want)[i] + 4 * 2.0 ** -15 * s[i]
    to_file(bad)
    assert not files.verify_result("output/output.bin", "output/golden.bin", mode="fp8")
    # the old verifier (atol = 8 * 2^-15 * max|golden|, 1e-4 of the elements free) accepted this file
    assert 1 / bad.size <= 1e-4
    (tmp_path / "input" / "sfa.bin").unlink()
    assert not files.verify_result("output/output.bin", "output/golden.bin", mode="fp8")     # inputs missing: no S, no pass


def test_sweep_gate_uses_the_same_bar():
    torch = pytest.importorskip("torch")
    from deepgemm_ascend_amd.harness import sweep
    a, sfa, b, sfb, golden, s = _case()
    g, ss = torch.from_numpy(golden), torch.from_numpy(s).float()
    want = g.to(torch.bfloat16)
    ok, frac = sweep.is_correct(g, want, ss)
    assert ok and frac == 0.0
    bad = want.float().clone()
    bad[3, 3] += 8 * float(want[3, 3].abs()) * 2.0 ** -7 + 8 * 2.0 ** -15 * float(ss[3, 3])
    ok, frac = sweep.is_correct(g, bad.to(torch.bfloat16), ss)
    assert not ok
    with pytest.raises(ValueError):
        sweep.is_correct(g, want)
