"""policy="fast_ue8m0" (DGA_POLICY_UE8M0_SCALES): the block scales ride in the E8M0 operands of
v_mfma_scale_f32_16x16x128_f8f6f4 and the MFMA accumulates in place -- the CDNA4 reading of the reference's
Mmad(c1Local, ..., init on first) accumulate-in-place K loop (/root/reference/deep_gemm_ascend/framework/csrc/jit/generate_code.hpp:320-335).
For scale tensors whose values are exact powers of two (upstream DeepGEMM's use_ue8m0 quantisation).  Bar: the fast policy's
(oracle.assert_parity against the CPU oracle), and -- the statement that lets a caller switch -- the same bf16 outputs as
policy="fast" on the same inputs (fp32 rounding order is the only difference: profiles/r05_probe_scale_acc.txt)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _pow2_ceil(s):
    return np.exp2(np.ceil(np.log2(s))).astype(np.float32)


def _inputs(m, n, k, seed):
    """8(d) recipe with the scales rounded UP to powers of two (2^ceil(log2(amax / 448))), quantised by torch on the CPU."""
    rng = np.random.default_rng(seed)
    kb, nb = (k + 127) // 128, (n + 127) // 128
    xa = np.zeros((m, kb * 128), np.float32); xa[:, :k] = rng.standard_normal((m, k), dtype=np.float32)
    xb = np.zeros((nb * 128, kb * 128), np.float32); xb[:n, :k] = rng.standard_normal((n, k), dtype=np.float32)
    sfa = _pow2_ceil(np.maximum(np.abs(xa).reshape(m, kb, 128).max(axis=2), 1e-30) / 448.0)
    sfb = _pow2_ceil(np.maximum(np.abs(xb).reshape(nb, 128, kb, 128).max(axis=(1, 3)), 1e-30) / 448.0)
    qa = torch.from_numpy(xa.reshape(m, kb, 128) / sfa[:, :, None]).reshape(m, kb * 128).to(torch.float8_e4m3fn).view(torch.uint8).numpy()[:, :k]
    qb = torch.from_numpy(xb.reshape(nb, 128, kb, 128) / sfb[:, None, :, None]).reshape(nb * 128, kb * 128).to(torch.float8_e4m3fn).view(torch.uint8).numpy()[:n, :k]
    return np.ascontiguousarray(qa), sfa, np.ascontiguousarray(qb), sfb


def _run(dga, a, sfa, b, sfb, policy, tiling_=None):
    out = torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()), (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()),
                             out, policy=policy, tiling_=tiling_, sync=True)
    return _bits(out)


@pytest.mark.parametrize("m,n,k", [
    (256, 256, 512),        # one 256 x 256 tile
    (512, 768, 1024),       # the continuous loop over several tiles
    (300, 520, 1040),       # ragged in every axis, K tail chunk
    (1, 128, 128), (17, 130, 144), (129, 257, 4096), (64, 4096, 2048),
    (1024, 2048, 896),      # loader-wave tiles
    (200, 392, 1921),       # odd K: padding pass, then the tile kernel
    (16, 4096, 7168),       # decode: a kernel without a hardware-scale build (the promotion form answers)
])
def test_parity_against_the_oracle_and_the_promotion_form(dga, oracle, m, n, k):
    a, sfa, b, sfb = _inputs(m, n, k, seed=m + n + k)
    got = _run(dga, a, sfa, b, sfb, "fast_ue8m0")
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    oracle.assert_parity(got, want, a, sfa, b, sfb)
    ref = _run(dga, a, sfa, b, sfb, "fast")
    d = oracle.bf16_ulp_diff(got, ref)
    assert int(d.max(initial=0)) <= 1 and float((d > 0).mean()) <= 1e-5, (int(d.max(initial=0)), float((d > 0).mean()))


@pytest.mark.parametrize("tile,waves,stages,tag", [
    ((256, 256), (4, 2), 2, 2), ((256, 256), (4, 2), 2, 0), ((256, 256), (4, 2), 2, 6),
    ((128, 256), (2, 2), 3, 4), ((128, 256), (2, 2), 3, 0), ((128, 256), (2, 2), 3, 5),
    ((128, 128), (2, 2), 3, 4), ((128, 128), (2, 2), 3, 0), ((64, 256), (1, 4), 3, 4), ((64, 256), (1, 4), 3, 0),
    ((64, 128), (1, 4), 3, 4), ((64, 128), (1, 4), 3, 0),
    ((32, 128), (1, 4), 3, 0),       # no hardware-scale build: the promotion form runs, the flag is a promise, not a demand
])
def test_every_hardware_scale_build_by_name(dga, oracle, tile, waves, stages, tag):
    m, n, k = 520, 1030, 1152
    a, sfa, b, sfb = _inputs(m, n, k, seed=tile[0] + tile[1] + tag)
    t = dga.tiling(m, n, k)
    t.m1, t.n1 = tile; t.wavesM, t.wavesN = waves; t.stages = stages; t.dispatchPolicyTag = tag | 16
    t.splitkFactor = 1; t.kernelSerial = 0
    assert dga.tiling_check(t) == 0
    got = _run(dga, a, sfa, b, sfb, None, tiling_=t)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    oracle.assert_parity(got, want, a, sfa, b, sfb)
    t.dispatchPolicyTag = tag
    ref = _run(dga, a, sfa, b, sfb, None, tiling_=t)
    d = oracle.bf16_ulp_diff(got, ref)
    assert int(d.max(initial=0)) <= 1 and float((d > 0).mean()) <= 1e-5


def test_split_k_and_quarter_tile_tail_with_the_flag(dga, oracle):
    a, sfa, b, sfb = _inputs(96, 640, 4096, seed=4)
    t = dga.tiling(96, 640, 4096)
    t.splitkFactor = 4; t.kernelSerial = 4; t.dispatchPolicyTag = (t.dispatchPolicyTag & 7) | 16
    got = _run(dga, a, sfa, b, sfb, None, tiling_=t)
    oracle.assert_parity(got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)
    # 17 x 16 = 272 tiles of 256 x 256: one round of 256 + a 16-tile tail in quarter tiles
    m, n, k = 17 * 256, 16 * 256, 256
    a, sfa, b, sfb = _inputs(m, n, k, seed=6)
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages = 256, 256, 4, 2, 2
    t.kernelSerial = 5; t.splitkFactor = 1; t.dispatchPolicyTag = 2 | 16
    got = _run(dga, a, sfa, b, sfb, None, tiling_=t)
    rows = np.r_[0:64, m - 300:m]           # the first round's rows and the tail's
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    oracle.assert_parity(got[rows], want, a[rows], sfa[rows], b, sfb)


def test_masked_grouped_and_contiguous_layouts(dga, oracle):
    g, mmax, n, k = 6, 128, 512, 1024
    parts = [_inputs(mmax, n, k, seed=50 + i) for i in range(g)]
    A, SFA, B, SFB = (np.stack([p[j] for p in parts]) for j in range(4))
    masked = np.array([0, 1, 77, 128, 64, 127], np.int32)
    dev = lambda x: torch.from_numpy(x).cuda()
    outs = {}
    for pol in ("fast_ue8m0", "fast"):
        out = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((dev(A), dev(SFA)), (dev(B), dev(SFB)), out, dev(masked), expected_m=96, policy=pol, sync=True)
        outs[pol] = _bits(out)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(A, SFA, B, SFB, np.zeros((g, mmax, n), np.uint16), masked)
    for i in range(g):
        mm = int(masked[i])
        assert (outs["fast_ue8m0"][i, mm:] == 0).all(), "rows >= masked_m were written"
        if mm:
            oracle.assert_parity(outs["fast_ue8m0"][i, :mm], want[i, :mm], A[i, :mm], SFA[i, :mm], B[i], SFB[i])
    d = oracle.bf16_ulp_diff(outs["fast_ue8m0"], outs["fast"])
    assert int(d.max(initial=0)) <= 1 and float((d > 0).mean()) <= 1e-5
    # contiguous layout: three 128-row segments
    idx = np.repeat(np.array([2, 0, 5], np.int32), 128)
    a2 = np.concatenate([A[2], A[0], A[5]]); s2 = np.concatenate([SFA[2], SFA[0], SFA[5]])
    out = torch.zeros((384, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((dev(a2), dev(s2)), (dev(B), dev(SFB)), out, dev(idx), policy="fast_ue8m0", sync=True)
    got = _bits(out)
    for j, gi in enumerate((2, 0, 5)):
        w = oracle.gemm_fp8_fp8_bf16_nt(A[gi], SFA[gi], B[gi], SFB[gi], threads=8)
        oracle.assert_parity(got[128 * j:128 * (j + 1)], w, A[gi], SFA[gi], B[gi], SFB[gi])


def test_a_scale_that_is_not_a_power_of_two_is_read_as_its_exponent(dga, oracle):
    """The stated behaviour of a broken promise (include/dga_hip.h): the mantissa of the scale is dropped."""
    m, n, k = 256, 256, 256
    a, sfa, b, sfb = _inputs(m, n, k, seed=9)
    sfa_bad = (sfa * np.float32(1.75)).astype(np.float32)
    sfb_bad = (sfb * np.float32(1.25)).astype(np.float32)
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = 256, 256, 4, 2, 2, 2 | 16
    t.kernelSerial = 0; t.splitkFactor = 1
    got = _run(dga, a, sfa_bad, b, sfb_bad, None, tiling_=t)
    ref = _run(dga, a, sfa, b, sfb, None, tiling_=t)     # floor to the power of two below = the original scales
    assert np.array_equal(got, ref)


def test_policy_name_rules(dga):
    t = dga.tiling(512, 512, 512, policy="bf16_exact")
    a = torch.zeros((512, 512), dtype=torch.uint8, device="cuda")
    s = torch.ones((512, 4), device="cuda"); sb = torch.ones((4, 4), device="cuda")
    out = torch.empty((512, 512), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(Exception):
        dga.gemm_fp8_fp8_bf16_nt((a, s), (a, sb), out, policy="fast_ue8m0", tiling_=t)       # a bf16-exact tiling is not a fast-path tiling
    with pytest.raises(Exception):
        dga.gemm_fp8_fp8_bf16_nt((a, s), (a, sb), out, policy="fast_ue8m0", strict=True)


# ---- policy "bf16_exact_ue8m0" (7 | 16): the in-contract arithmetic with the power-of-two scales folded into the exact e4m3 -> bf16
#      conversions, the bf16 MFMA chain accumulating in place (gemm_fp8_kernel.hpp MATH = 3).  Held to the bf16-exact policy's bar.

def _assert_in_contract(oracle, got, want, a, sfa, b, sfb):
    rep = oracle.parity_report(got, want, a, sfa, b, sfb)
    size = int(np.asarray(got).size)
    assert rep["nan_positions_equal"]
    assert int(round(rep["frac_gt_max_ulp"] * size)) <= max(1e-5 * size, 2), rep
    assert rep["worst_excess_over_S"] <= 2.0 ** -22, rep


@pytest.mark.parametrize("m,n,k", [(128, 256, 512), (300, 520, 1040), (512, 768, 1024), (129, 257, 4096), (64, 512, 2048), (200, 392, 1921),
                                   (1024, 2048, 896)])
def test_bf16_exact_ue8m0_meets_the_contract(dga, oracle, m, n, k):
    a, sfa, b, sfb = _inputs(m, n, k, seed=m + 3 * n + k)
    got = _run(dga, a, sfa, b, sfb, "bf16_exact_ue8m0")
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    _assert_in_contract(oracle, got, want, a, sfa, b, sfb)


@pytest.mark.parametrize("tile", [(128, 256), (64, 256), (128, 128), (32, 128)])
def test_bf16_exact_ue8m0_builds_by_name(dga, oracle, tile):
    """128 x 256 and 64 x 256 have a build with the scales folded in; the other tiles run the bf16-exact policy as it is."""
    m, n, k = 520, 1030, 1152
    a, sfa, b, sfb = _inputs(m, n, k, seed=tile[0] + tile[1])
    t = dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1 = tile; t.wavesM = t.wavesN = 0; t.stages = 3; t.splitkFactor = 1; t.kernelSerial = 0; t.dispatchPolicyTag = 7 | 16
    assert dga.tiling_check(t) == 0
    got = _run(dga, a, sfa, b, sfb, None, tiling_=t)
    _assert_in_contract(oracle, got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)
    t.splitkFactor = 3; t.kernelSerial = 4           # split-K slabs of the same builds
    got = _run(dga, a, sfa, b, sfb, None, tiling_=t)
    _assert_in_contract(oracle, got, oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8), a, sfa, b, sfb)


def test_bf16_exact_ue8m0_masked_grouped(dga, oracle):
    g, mmax, n, k = 6, 128, 512, 1024
    parts = [_inputs(mmax, n, k, seed=70 + i) for i in range(g)]
    A, SFA, B, SFB = (np.stack([p[j] for p in parts]) for j in range(4))
    masked = np.array([0, 1, 77, 128, 64, 127], np.int32)
    dev = lambda x: torch.from_numpy(x).cuda()
    out = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((dev(A), dev(SFA)), (dev(B), dev(SFB)), out, dev(masked), expected_m=96,
                                              policy="bf16_exact_ue8m0", sync=True)
    got = _bits(out)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(A, SFA, B, SFB, np.zeros((g, mmax, n), np.uint16), masked)
    for i in range(g):
        mm = int(masked[i])
        assert (got[i, mm:] == 0).all(), "rows >= masked_m were written"
        if mm:
            _assert_in_contract(oracle, got[i, :mm], want[i, :mm], A[i, :mm], SFA[i, :mm], B[i], SFB[i])
