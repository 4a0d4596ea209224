"""The one-launch workgroup split-K kernel for short-M problems (csrc/gemm_fp8_wsk_kernel.hpp, kernelSerial 6): the 8 waves of a
workgroup are the 8 K slices of one output tile, partial tiles combined in LDS.  Two builds: fragments streamed global ->
registers (M <= 64; a tiling with build = 1, DGA_BUILD_WSK_REGISTER, names it), and -- M <= 32, the default there -- operands staged through per-wave
LDS-DMA rings (whole-line requests, hand-counted vmcnt).

Reference counterparts: the Stream-K kernel's fused reduce
(/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/kernel/padding_streamk_matmul_kernel.h:92-107) and the single-core
split-K kernel types of op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36.

Bars: (1) BIT IDENTITY with the two-launch split-K (tile kernel writing fp32 slabs + combine kernel) at splitkFactor 8 -- the
same K slices, the same per-slice arithmetic, the same combine order; (2) the fast path's bar against the CPU oracle.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _run(dga, a, sfa, b, sfb, wsk, split=8, policy=None):
    """wsk: False = two-launch split-K `split`; True / "dma" = kernelSerial 6 as the dispatcher builds it (8-wave LDS-DMA rings up to
    32 rows); "reg" = the register build"""
    m, k = a.shape
    n = b.shape[0]
    t = dga.tiling(m, n, k)
    if wsk:
        t.kernelSerial, t.splitkFactor = 6, 1
        t.stages, t.build = 3, (1 if wsk == "reg" else 0)
    else:
        t.kernelSerial, t.splitkFactor = 4, split
        t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.dispatchPolicyTag = 64, 128, 3, 1, 4, 0
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True, tiling_=t, policy=policy)
    return _bits(out)


@pytest.mark.parametrize("m,n,k", [
    (1, 80, 1024), (8, 512, 2048), (16, 1000, 4096 + 16), (17, 333, 1040), (32, 4096, 1024), (33, 640, 3072),
    (64, 1024, 1152), (64, 256, 128),      # one k block: seven of the eight waves have nothing to do
    (5, 48, 16), (48, 7168, 7168), (13, 72, 8192 + 48),
    (16, 16 * 256 + 16, 2048),            # 257 n-tiles: one workgroup walks two
    (9, 16 * 700 + 5, 1024 + 32),         # 2.7 n-tiles per workgroup: passes of 2 + 1, the last column tile cut
    (32, 16 * 600, 1152), (24, 40, 128 * 9),
    (64, 16 * 300, 2048 + 16), (40, 16 * 1300 + 3, 1024), (20, 16 * 900, 640), (30, 16 * 1600, 512),
])
def test_against_two_launch_split_k_and_the_oracle(dga, oracle, m, n, k):
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + n + k)
    da, dsfa, db, dsfb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
    got = _run(dga, da, dsfa, db, dsfb, "dma")
    ref = _run(dga, da, dsfa, db, dsfb, False)
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ from the two-launch split-K"
    got_reg = _run(dga, da, dsfa, db, dsfb, "reg")
    assert np.array_equal(got_reg, ref), f"register build: {int((got_reg != ref).sum())} of {got.size} outputs differ"
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    oracle.assert_parity(got, want, a, sfa, b, sfb)


@pytest.mark.parametrize("m,n,k", [(8, 18432, 7168), (64, 4096, 7168), (16, 7168, 18432), (64, 18432, 7168)])
def test_decode_shapes_at_full_size(dga, m, n, k):
    """The decode rows of the reference's shape list (framework/benchmark/benchmark.py:24-44) at full size: every output
    bit-identical to the two-launch split-K."""
    gen = torch.Generator(device="cuda").manual_seed(m * n + k)
    a = torch.randint(0, 256, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 256, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
    a[(a & 0x7F) == 0x7F] = 0x3C; b[(b & 0x7F) == 0x7F] = 0x3C        # no NaN codes
    kb = (k + 127) // 128
    sfa = torch.rand((m, kb), device="cuda", generator=gen) + 0.5
    sfb = torch.rand(((n + 127) // 128, kb), device="cuda", generator=gen) + 0.5
    got = _run(dga, a, sfa, b, sfb, True)
    ref = _run(dga, a, sfa, b, sfb, False)
    assert np.array_equal(got, ref)
    assert np.array_equal(_run(dga, a, sfa, b, sfb, "reg"), ref)


@pytest.mark.parametrize("m,n,k", [
    (1, 80, 1024), (8, 512, 2048), (16, 1000, 4096 + 16), (17, 333, 1040), (32, 4096, 1024), (5, 48, 16),
    (16, 16 * 256 + 16, 2048), (9, 16 * 700 + 5, 1024 + 32), (32, 16 * 600, 1152), (13, 72, 8192 + 48),
])
def test_bf16_exact_policy_on_the_same_rings(dga, oracle, m, n, k):
    """dispatchPolicyTag 7 with kernelSerial 6 (M <= 32): bit-identical to that policy's two-launch split-K 8, and inside the
    policy's bar against the oracle."""
    from test_bf16_exact_gpu import _assert_bar, EPS, EPS_ARBITRARY
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=2 * m + n + k)
    da, dsfa, db, dsfb = _dev(a), _dev(sfa), _dev(b), _dev(sfb)
    got = _run(dga, da, dsfa, db, dsfb, "dma", policy="bf16_exact")
    ref = _run(dga, da, dsfa, db, dsfb, False, policy="bf16_exact")
    assert np.array_equal(got, ref), f"{int((got != ref).sum())} of {got.size} outputs differ from the two-launch split-K"
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    _assert_bar(oracle, got, want, a, sfa, b, sfb, eps=EPS if k >= 128 else EPS_ARBITRARY)


def test_policy_auto_is_exact_where_the_decode_kernel_carries_it(dga, oracle):
    """policy="auto": the bf16-exact arithmetic on the rows the workgroup split-K takes, the fast policy elsewhere (same bytes as
    naming the policy)."""
    for (m, n, k), exact in (((8, 512, 2048), True), ((16, 4096, 7168), True), ((256, 512, 1024), False), ((64, 18432, 1024), False)):
        a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=m + k)
        ops = ((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)))
        outs = {}
        for pol in ("auto", "bf16_exact", "fast"):
            out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
            dga.gemm_fp8_fp8_bf16_nt(*ops, out, sync=True, policy=pol)
            outs[pol] = _bits(out)
        assert np.array_equal(outs["auto"], outs["bf16_exact" if exact else "fast"]), (m, n, k)


def test_shapes_it_does_not_take_fall_through(dga, oracle):
    """M > 64 with kernelSerial 6 runs the tiling's tile kernel (same answer as without the request)."""
    m, n, k = 96, 256, 512
    a, sfa, b, sfb = oracle.make_inputs(m, n, k, seed=3)
    t = dga.tiling(m, n, k)
    t.kernelSerial = 6
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((_dev(a), _dev(sfa)), (_dev(b), _dev(sfb)), out, sync=True, tiling_=t)
    oracle.assert_parity(_bits(out), oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=4), a, sfa, b, sfb)
