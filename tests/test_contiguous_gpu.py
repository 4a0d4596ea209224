"""GPU parity of the contiguous-grouped (prefill MoE) layout against the oracle: row r uses b[m_indices[r]], rows with
index -1 are padding and stay untouched.  No reference counterpart; the definition of record
is oracle.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ALIGN = 128
SENTINEL = np.uint16(0x7FC1)


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _layout(counts, align=ALIGN):
    """m_indices for groups with `counts` valid rows each: segments padded with -1 to a multiple of `align`."""
    idx = []
    for g, c in enumerate(counts):
        seg = -(-c // align) * align
        idx += [g] * c + [-1] * (seg - c)
    return np.array(idx, np.int32)


def _inputs(oracle, counts, n, k, seed, groups=None):
    m_indices = _layout(counts)
    g = groups or len(counts)
    msum = m_indices.size
    a, sfa, _, _ = oracle.make_inputs(msum, 8, k, seed=seed)
    B, SFB = [], []
    for i in range(g):
        _, _, b, sfb = oracle.make_inputs(8, n, k, seed=seed * 100 + i + 1)
        B.append(b); SFB.append(sfb)
    return a, sfa, np.stack(B), np.stack(SFB), m_indices


def _run(dga, a, sfa, b, sfb, m_indices, n, tiling_=None):
    init = np.full((a.shape[0], n), SENTINEL, np.uint16)
    out = torch.from_numpy(init.view(np.int16)).cuda().view(torch.bfloat16)
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(
        (torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
        (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()),
        out, torch.from_numpy(m_indices).cuda(), tiling_=tiling_, sync=True)
    return init, _bits(out)


@pytest.mark.parametrize("counts,n,k", [
    ([128, 128, 128], 256, 512),        # full segments
    ([100, 1, 0, 200, 128], 384, 384),  # ragged segments, an empty group, a two-block group
    ([37], 136, 144),                   # one group, N tail, K tail chunk (K % 128 != 0)
    ([40, 250, 64], 128, 1921),         # odd K: padding pass (workspace) route
])
def test_contiguous_parity(dga, oracle, counts, n, k):
    a, sfa, b, sfb, m_indices = _inputs(oracle, counts, n, k, seed=len(counts) + n)
    init, got = _run(dga, a, sfa, b, sfb, m_indices, n)
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a, sfa, b, sfb, init, m_indices, threads=8)
    pad = m_indices < 0
    assert (got[pad] == SENTINEL).all(), "padding rows were written"
    for g in range(b.shape[0]):
        rows = np.nonzero(m_indices == g)[0]
        if rows.size:
            oracle.assert_parity(got[rows], want[rows], a[rows], sfa[rows], b[g], sfb[g])


@pytest.mark.parametrize("bm", [16, 32, 64, 128, 256])
def test_contiguous_every_tile_height(dga, oracle, bm):
    """Every tile height that divides the alignment gives the same bytes (the group lookup is per tile)."""
    counts, n, k = [70, 128, 3], 256, 256
    a, sfa, b, sfb, m_indices = _inputs(oracle, counts, n, k, seed=9)
    t = dga.tiling(m_indices.size, n, k, groups=len(counts), contiguous=True)
    _, base = _run(dga, a, sfa, b, sfb, m_indices, n, tiling_=t)
    t2 = dga.tiling(m_indices.size, n, k, groups=len(counts), contiguous=True)
    t2.m1 = bm; t2.n1 = 256; t2.wavesM = 0; t2.wavesN = 0; t2.stages = 2
    _, got = _run(dga, a, sfa, b, sfb, m_indices, n, tiling_=t2)
    assert (got == base).all()


@pytest.mark.parametrize("counts", [
    [384, 384, 200, 700],       # group boundaries at odd multiples of 128: every other 256-row tile straddles two groups
    [128, 128, 128, 128, 128],  # every tile straddles
    [0, 300, 0, 1, 512],        # empty groups, a padding-only second block, a one-row group
])
def test_contiguous_two_pass_tiles(dga, oracle, counts):
    """256-row tiles (taller than the alignment): pass 0 / pass 1 of a straddling tile each store their own group's
    rows; parity against the oracle and bitwise equality with the 128-row tiling."""
    n, k = 256, 384
    a, sfa, b, sfb, m_indices = _inputs(oracle, counts, n, k, seed=sum(counts))
    t = dga.tiling(m_indices.size, n, k, groups=len(counts), contiguous=True)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = 128, 256, 0, 0, 2, 0
    _, base = _run(dga, a, sfa, b, sfb, m_indices, n, tiling_=t)
    for policy in (0, 2):
        t2 = dga.tiling(m_indices.size, n, k, groups=len(counts), contiguous=True)
        t2.m1, t2.n1, t2.wavesM, t2.wavesN, t2.stages, t2.dispatchPolicyTag = 256, 256, 0, 0, 2, policy
        init, got = _run(dga, a, sfa, b, sfb, m_indices, n, tiling_=t2)
        assert (got[m_indices < 0] == SENTINEL).all()
        assert (got == base).all()
    want = oracle.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a, sfa, b, sfb, init, m_indices, threads=8)
    for g in range(b.shape[0]):
        rows = np.nonzero(m_indices == g)[0]
        if rows.size >= 32:
            oracle.assert_parity(got[rows], want[rows], a[rows], sfa[rows], b[g], sfb[g])


def test_contiguous_rejects_tile_heights_outside_the_menu(dga, oracle):
    counts, n, k = [128, 128], 256, 256
    a, sfa, b, sfb, m_indices = _inputs(oracle, counts, n, k, seed=3)
    t = dga.tiling(256, n, k, groups=2, contiguous=True)
    t.m1 = 96  # does not divide the 128-row alignment
    with pytest.raises(dga.DGAError):
        _run(dga, a, sfa, b, sfb, m_indices, n, tiling_=t)


def test_contiguous_equals_masked_layout(dga, oracle):
    """Size-independent property at a larger shape: the contiguous layout with full 128-row segments gives the same
    bytes as the masked layout on the same operands."""
    g, n, k = 16, 1024, 2048
    a, sfa, b, sfb, m_indices = _inputs(oracle, [128] * g, n, k, seed=21)
    _, got = _run(dga, a, sfa, b, sfb, m_indices, n)
    ta, tsfa, tb, tsfb = [torch.from_numpy(x).cuda() for x in (a, sfa, b, sfb)]
    out = torch.zeros((g, 128, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((ta.view(g, 128, k), tsfa.view(g, 128, -1)), (tb, tsfb), out,
                                              torch.full((g,), 128, dtype=torch.int32, device="cuda"), expected_m=128,
                                              sync=True)
    assert (got.reshape(g, 128, n) == _bits(out)).all()


def test_contiguous_bad_index_is_ignored(dga, oracle):
    """An index >= G must not read outside b: the tile is skipped (rows untouched)."""
    counts, n, k = [128, 128], 128, 128
    a, sfa, b, sfb, m_indices = _inputs(oracle, counts, n, k, seed=5)
    bad = m_indices.copy(); bad[128:] = 7
    _, got = _run(dga, a, sfa, b, sfb, bad, n)
    assert (got[128:] == SENTINEL).all() and (got[:128] != SENTINEL).any()


def test_contiguous_degenerate_shapes(dga, oracle):
    """m_sum = 0, G = 0 and K = 0 (valid rows become zeros, padding rows stay untouched)."""
    e = lambda *s: torch.empty(s, dtype=torch.uint8, device="cuda")
    f = lambda *s: torch.empty(s, dtype=torch.float32, device="cuda")
    out = torch.empty((0, 128), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((e(0, 256), f(0, 2)), (e(2, 128, 256), f(2, 1, 2)), out,
                                                  torch.empty((0,), dtype=torch.int32, device="cuda"), sync=True)
    idx = torch.tensor([0] * 100 + [-1] * 28, dtype=torch.int32, device="cuda")
    init = np.full((128, 128), SENTINEL, np.uint16)
    out = torch.from_numpy(init.view(np.int16)).cuda().view(torch.bfloat16)
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((e(128, 0), f(128, 0)), (e(1, 128, 0), f(1, 1, 0)), out, idx, sync=True)
    got = _bits(out)
    assert (got[:100] == 0).all() and (got[100:] == SENTINEL).all()
    out2 = torch.from_numpy(init.view(np.int16)).cuda().view(torch.bfloat16)
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((e(128, 256), f(128, 2)), (e(0, 128, 256), f(0, 1, 2)), out2, idx, sync=True)
    assert (_bits(out2) == SENTINEL).all()
