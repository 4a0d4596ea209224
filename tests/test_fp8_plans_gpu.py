"""Any tiling the fp8 launcher ACCEPTS computes the product: every build of the menu (tile, waves, stages) x every dispatch policy x
split-K 1 / 2 / 3 forced on one ragged problem through the C ABI (dga_gemm_fp8_fp8_bf16_nt with a caller's tiling -- the reference's
`TilingParams` handed to the device entry, /root/reference/aclnn_catlass_dynamic_matmul/op_host/op_tiling/tiling_params.h:19-66),
checked against the CPU oracle under the fast path's bar.  A combination the menu does not hold is mapped onto a build it does hold
(or refused with DGA_E_TILING): whichever the launcher does, the bytes must be the product's.  (The 16-bit paths' counterpart, tests/test_op16_plans_gpu.py, found a split-K slice
ignored by one loop; this is the same net under the fp8 menu.)"""
import numpy as np
import pytest
import torch

from deepgemm_ascend_amd.harness import sweep

pytestmark = pytest.mark.gpu

M, N, K = 300, 520, 1344      # ragged in every dimension of every tile; 11 k blocks with a 64-wide tail: uneven slices


@pytest.fixture(scope="module")
def problem(oracle):
    a, sfa, b, sfb = oracle.make_inputs(M, N, K, seed=11)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    dev = tuple(torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (a, sfa, b, sfb))
    return (a, sfa, b, sfb), dev, want


TALLY = {"ran": 0, "refused": 0}
CASES = [(bm, bn, wm, wn, st, pol, sk) for (bm, bn, wm, wn, st, _) in sweep.MENU for pol in (0, 1, 2, 4, 5, 6, 7) for sk in (1, 2, 3)]


@pytest.mark.parametrize("bm,bn,wm,wn,st,pol,sk", CASES)
def test_accepted_tilings_are_correct(dga, oracle, problem, bm, bn, wm, wn, st, pol, sk):
    host, dev, want = problem
    t = dga.tiling(M, N, K)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = bm, bn, wm, wn, st, pol
    t.splitkFactor, t.kernelSerial = sk, (4 if sk > 1 else 0)
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    try:
        dga.gemm_fp8_fp8_bf16_nt((dev[0], dev[1]), (dev[2], dev[3]), out, tiling_=t, sync=True)
    except RuntimeError as e:
        assert "TILING" in str(e).upper() or "tiling" in str(e), e     # refused: fine
        TALLY["refused"] += 1
        return
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    oracle.assert_parity(got, want, *host)
    TALLY["ran"] += 1


def test_the_net_is_not_empty():
    if TALLY["ran"] + TALLY["refused"] < len(CASES):
        pytest.skip("run together with the cases above")
    print("fp8 plans:", TALLY)
    assert TALLY["ran"] >= 100, TALLY


KS_CASES = [(m, n, k, bm, bn, st, ks) for (m, n, k) in ((300, 520, 1344), (40, 1000, 2048), (70, 333, 1001))
            for (bm, bn, st) in ((256, 256, 2), (128, 256, 3), (64, 128, 3), (16, 128, 3), (16, 128, 1))
            for ks in (0, 1, 2, 4, 5, 6)]


@pytest.mark.parametrize("m,n,k,bm,bn,st,ks", KS_CASES)
def test_any_kernel_serial_a_caller_names(dga, oracle, m, n, k, bm, bn, st, ks):
    """kernelSerial is the caller's too (the reference's TilingParams carries it, tiling_params.h:19-66): every value of the menu named
    on every kind of problem -- ragged, decode-sized, odd K -- with tiles that do and do not have the kernel in question.  The launcher
    runs the kernel where it applies and the tiling's tile kernel where it does not; the bytes are the product's either way."""
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + k)
    dev = tuple(torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (a, sfa, b, sfb))
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = bm, bn, 0, 0, st, 0
    t.splitkFactor, t.kernelSerial = (2 if ks == 4 else 1), ks
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    try:
        dga.gemm_fp8_fp8_bf16_nt((dev[0], dev[1]), (dev[2], dev[3]), out, tiling_=t, sync=True)
    except RuntimeError as e:
        assert "tiling" in str(e).lower(), e
        return
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    if m * n >= 2048 * 8:
        oracle.assert_parity(got, want, a, sfa, b, sfb)
    else:
        rep = oracle.parity_report(got, want, a, sfa, b, sfb)
        assert rep["nan_positions_equal"] and rep["worst_excess_over_S"] <= oracle.eps_for_k(k), rep
