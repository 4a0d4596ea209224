"""The bf16-exact policy's 128 x 256 tile with its last partial round in quarter tiles (kernelSerial 5, csrc/dga_launch.hip: the whole
rounds in one launch, the remaining tiles as 64 x 128 tiles, four per parent tile, in a second one) against the single launch of the
same tile: the same arithmetic in the same k order, so the bar is BIT IDENTITY; against the oracle it is the policy's bar
(tests/test_bf16_exact_gpu.py).  Counterpart in the reference: the Stream-K handler of its kernel selector, which exists to fill the
last partial wave of cores (/root/reference/aclnn_catlass_dynamic_matmul/op_host/select_kernel.cpp:303-331)."""
import numpy as np
import pytest
import torch

from test_bf16_exact_gpu import _assert_bar, _bits, _dev, EPS

pytestmark = pytest.mark.gpu


def _tiling(dga, m, n, k, tail):
    t = dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1, t.splitkFactor, t.dispatchPolicyTag, t.wavesM, t.wavesN = 128, 256, 1, 7, 0, 0
    t.kernelSerial, t.stages, t.build = (5, 3, 0) if tail else (0, 3, 8)
    return t


def _run(dga, a, sfa, b, sfb, t):
    out = torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, policy="bf16_exact", sync=True, tiling_=t)
    return _bits(out)


@pytest.mark.parametrize("m,n,k", [
    (2304, 4096, 256),          # 288 tiles: 256 + a tail of 32 = 128 quarter tiles
    (2250, 4100, 272),          # 306 tiles with every edge cut (rows, columns, K % 128 = 16): quarter tiles beyond both matrix edges
    (3072, 4096, 128),          # 384 tiles: the longest tail the launcher takes (128 = half the CUs, two quarter tiles to a CU)
    (1300, 16000, 144),         # 693 tiles, tail 181: longer than half the CUs -- the launcher runs the single launch
    (128, 512, 384),            # 2 tiles: no whole round -- the single launch
])
def test_tail_is_bit_identical_to_the_single_launch(dga, oracle, m, n, k):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + 3 * n + 7 * k)
    assert dga.tiling_check(_tiling(dga, m, n, k, True)) == 0
    got = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, True))
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, False))
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ"
    rows = np.r_[0:48, max(m - 300, 0):m]   # the head and the rows the tail tiles cover
    want = oracle.gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b, sfb, threads=8)
    _assert_bar(oracle, got[rows], want, a[rows], sfa[rows], b, sfb, eps=EPS)


def test_the_selector_names_the_tail_and_the_default_call_runs_it(dga, oracle):
    """2304 x 4096 x 7168 -- 288 tiles of 128 x 256, 1.125 rounds -- is a shape the policy's own tiling answers with kernelSerial 5; the
    operator's default call (no tiling, no policy) then writes the bytes of the single launch."""
    m, n, k = 2304, 4096, 7168
    t = dga.tiling(m, n, k, policy="bf16_exact")
    assert (t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.blockDim) == (128, 256, 5, 1, 256 + 4 * 32), t.as_dict()
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=11)
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, sync=True)
    ref = _run(dga, a, sfa, b, sfb, _tiling(dga, m, n, k, False))
    assert np.array_equal(_bits(out), ref)


def test_tail_on_another_tile_is_refused(dga):
    """kernelSerial 5 under this policy names the 128 x 256 tile's launch pair only: dga_tiling_check (and every entry) refuses the rest."""
    t = _tiling(dga, 2304, 4096, 256, True)
    t.m1, t.n1 = 64, 128
    assert dga.tiling_check(t) != 0
    t = _tiling(dga, 2304, 4096, 256, True)
    t.kernelSerial = 7          # (the one-launch Stream-K is a build of the 128 x 256 tile too since round 6: tests/test_bf16x_streamk_gpu.py)
    assert dga.tiling_check(t) == 0
    t.m1, t.n1 = 128, 128
    assert dga.tiling_check(t) != 0
