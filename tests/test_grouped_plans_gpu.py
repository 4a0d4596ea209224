"""The masked and contiguous grouped entries under every tiling a caller can hand them: each build of the menu x each dispatch policy
forced through the C ABI on one ragged problem, against the CPU oracle (rows at or beyond masked_m untouched; padding rows of the
contiguous layout untouched).  Whatever build the launcher maps a tiling onto, the bytes must be the product's.  (The grouped
counterpart of tests/test_fp8_plans_gpu.py; no reference counterpart for the layouts themselves -- SURVEY.md 8(c), upstream DeepGEMM's
convention.)"""
import numpy as np
import pytest
import torch

from deepgemm_ascend_amd.harness import sweep

pytestmark = pytest.mark.gpu

G, MMAX, N, K = 3, 100, 392, 1040
MASKS = np.array([0, 37, 100], np.int32)
CASES = [(bm, bn, wm, wn, st, pol) for (bm, bn, wm, wn, st, _) in sweep.MENU for pol in (0, 2, 4, 5, 7)]


@pytest.fixture(scope="module")
def masked_problem(oracle):
    parts = [oracle.make_inputs(MMAX, N, K, seed=70 + i) for i in range(G)]
    a, sfa, b, sfb = (np.stack([p[j] for p in parts]) for j in range(4))
    init = np.full((G, MMAX, N), 0x7FC1, np.uint16)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_masked(a, sfa, b, sfb, init, MASKS, threads=8)
    dev = tuple(torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (a, sfa, b, sfb))
    return (a, sfa, b, sfb), dev, want


@pytest.mark.parametrize("bm,bn,wm,wn,st,pol", CASES)
def test_masked_accepted_tilings_are_correct(dga, oracle, masked_problem, bm, bn, wm, wn, st, pol):
    host, dev, want = masked_problem
    t = dga.tiling(MMAX, N, K, groups=G, expected_m=MMAX)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = bm, bn, wm, wn, st, pol
    out = torch.from_numpy(np.full((G, MMAX, N), 0x7FC1, np.uint16).view(np.int16)).cuda().view(torch.bfloat16)
    try:
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((dev[0], dev[1]), (dev[2], dev[3]), out, torch.from_numpy(MASKS).cuda(),
                                                  expected_m=MMAX, tiling_=t, sync=True)
    except RuntimeError as e:
        assert "tiling" in str(e).lower(), e
        return
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    for g in range(G):
        mm = int(MASKS[g])
        assert (got[g, mm:] == 0x7FC1).all(), f"group {g}: rows beyond masked_m written"
        if mm:
            oracle.assert_parity(got[g, :mm], want[g, :mm], host[0][g, :mm], host[1][g, :mm], host[2][g], host[3][g])


@pytest.fixture(scope="module")
def contiguous_problem(oracle):
    counts = [130, 0, 257]
    idx = []
    for gi, c in enumerate(counts):
        idx += [gi] * c + [-1] * (-(-c // 128) * 128 - c)
    idx = np.array(idx, np.int32)
    a, sfa, _, _ = oracle.make_inputs(idx.size, 8, K, seed=5)
    bs = [oracle.make_inputs(8, N, K, seed=90 + i)[2:] for i in range(len(counts))]
    b = np.stack([x[0] for x in bs]); sfb = np.stack([x[1] for x in bs])
    init = np.full((idx.size, N), 0x7FC1, np.uint16)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a, sfa, b, sfb, init, idx, threads=8)
    dev = tuple(torch.from_numpy(np.ascontiguousarray(x)).cuda() for x in (a, sfa, b, sfb, idx))
    return (a, sfa, b, sfb, idx), dev, want


@pytest.mark.parametrize("bm,bn,wm,wn,st,pol", CASES)
def test_contiguous_accepted_tilings_are_correct(dga, oracle, contiguous_problem, bm, bn, wm, wn, st, pol):
    host, dev, want = contiguous_problem
    a, sfa, b, sfb, idx = host
    t = dga.tiling(idx.size, N, K, groups=b.shape[0], contiguous=True)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = bm, bn, wm, wn, st, pol
    out = torch.from_numpy(np.full((idx.size, N), 0x7FC1, np.uint16).view(np.int16)).cuda().view(torch.bfloat16)
    try:
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((dev[0], dev[1]), (dev[2], dev[3]), out, dev[4], tiling_=t, sync=True)
    except RuntimeError as e:
        assert "tiling" in str(e).lower(), e
        return
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    pad = idx < 0
    assert (got[pad] == 0x7FC1).all(), "padding rows written"
    for g in range(b.shape[0]):
        rows = np.nonzero(idx == g)[0]
        if rows.size:
            oracle.assert_parity(got[rows], want[rows], a[rows], sfa[rows], b[g], sfb[g])
