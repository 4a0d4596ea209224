"""Weight matrices beyond 2 GB and 4 GB (a vocabulary projection of 140 000 - 280 000 rows): every address the kernels form past the
32-bit range -- tile bases, the per-pass bases of the workgroup split-K, buffer ranges -- against the fp32 matmul of the same values.
(The operands of BASELINE's configs are tens of megabytes; nothing else in the suite crosses 2^31 bytes.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,n,k", [(8, 140000, 15360), (8, 140000, 16384), (40, 140000, 16000)])   # workgroup split-K / table plan / deep tiles
def test_operator_16_bit_beyond_4_gb(dga, m, n, k):
    g = torch.Generator(device="cuda").manual_seed(1)
    x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    assert w.numel() * 2 > (1 << 32)
    o = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.catlass_dynamic_matmul(x, w.t(), o, sync=True)
    ref = x.float() @ w.float().t()
    assert bool(((o.float() - ref).abs() <= 2.0 ** -7 * ref.abs() + 2.0 ** -12 * (x.float().abs() @ w.float().abs().t())).all())


@pytest.mark.parametrize("m,n,k", [(8, 140000, 16384), (8, 280000, 16384), (64, 140000, 16384)])
def test_fp8_beyond_2_and_4_gb(dga, m, n, k):
    from deepgemm_ascend_amd.harness import sweep
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    assert b.numel() > (1 << 31)
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True)
    ok, diff = sweep.is_correct(golden, out, s_abs)
    assert ok and not bool(torch.isnan(out.float()).any()), diff


@pytest.mark.parametrize("m,stages,policy", [(8, 3, "fast"), (8, 1, "fast"), (24, 3, "fast"), (8, 3, "bf16_exact")])
def test_fp8_workgroup_split_k_beyond_2_gb_by_name(dga, m, stages, policy):
    """kernelSerial 6 NAMED on a weight matrix past 2^31 bytes (the selector keeps it to N <= 65536): the pass-by-pass and the
    continuous-ring builds (the register build with build = 1, DGA_BUILD_WSK_REGISTER) rebase the B descriptor at every pass's first row, so 32-bit
    offsets never leave a pass -- the last rows of the matrix are the ones that would be wrong."""
    from deepgemm_ascend_amd.harness import sweep
    n, k = 140000, 16384
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    assert b.numel() > (1 << 31)
    t = dga.tiling(m, n, k, policy="bf16_exact" if policy == "bf16_exact" else None)
    t.kernelSerial, t.m1, t.n1, t.splitkFactor, t.stages, t.wavesM, t.wavesN = 6, 16 if m <= 16 else 32, 128, 1, 3, 0, 0
    t.build = 1 if stages == 1 else 0
    assert dga.tiling_check(t) == 0
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t, sync=True)
    ok, diff = sweep.is_correct(golden, out, s_abs, policy=policy)
    assert ok and not bool(torch.isnan(out.float()).any()), diff
