"""GPU parity of the quantisers upstream of the GEMM (per-1x128 activations, per-128x128 weights): byte-exact e4m3fn
codes and bit-exact fp32 scales against the oracle's quantiser (oracle.quant_1x128 / quant_128x128)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(dga, oracle, x_t, fn, ofn):
    q, sf = fn(x_t)
    torch.cuda.synchronize()
    x = x_t.float().cpu().numpy()
    wq, wsf = ofn(x)
    gq = q.view(torch.uint8).cpu().numpy(); gsf = sf.cpu().numpy()
    assert gsf.shape == wsf.shape and gq.shape == wq.shape
    assert (gsf.view(np.uint32) == wsf.view(np.uint32)).all(), "scales differ"
    bad = np.nonzero(gq != wq)
    assert bad[0].size == 0, f"{bad[0].size} codes differ, first at {bad[0][0]},{bad[1][0]}: " \
                             f"{gq[bad][0]:#x} vs {wq[bad][0]:#x} for x={x[bad][0]!r}"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,k", [(64, 512), (3, 128), (130, 1000), (17, 77), (256, 7168)])
def test_per_token_cast(dga, oracle, dtype, rows, k):
    g = torch.Generator(device="cuda").manual_seed(rows * 7 + k)
    x = (torch.randn((rows, k), device="cuda", generator=g) * 3.0).to(dtype)
    _check(dga, oracle, x, dga.per_token_cast_to_fp8, oracle.quant_1x128)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,k", [(128, 128), (256, 512), (200, 1000), (5, 77), (1024, 2048)])
def test_per_block_cast(dga, oracle, dtype, rows, k):
    g = torch.Generator(device="cuda").manual_seed(rows * 5 + k)
    x = (torch.randn((rows, k), device="cuda", generator=g) * 0.02).to(dtype)
    _check(dga, oracle, x, dga.per_block_cast_to_fp8, oracle.quant_128x128)


def test_cast_edge_values(dga, oracle):
    """Zeros, an all-zero block (scale 1), subnormal results, ties, huge dynamic range, NaN, tiny amax."""
    x = np.zeros((8, 256), np.float32)
    x[0, :128] = np.linspace(-1, 1, 128)           # ordinary
    x[1, 0] = 448.0; x[1, 1:9] = [2.0 ** -9, 2.0 ** -10, 3 * 2.0 ** -10, 1e-3, -1e-3, 2.0 ** -6, 17.0, 19.0]
    x[2, :128] = 0.0                               # all-zero block -> scale 1, codes 0
    x[2, 128:] = -0.0
    x[3, :128] = np.float32(1e-38) * np.arange(128)  # tiny amax: the scale is subnormal-adjacent
    x[4, :128] = np.float32(3e38) * np.linspace(-1, 1, 128)
    x[5, :128] = np.arange(128) * 0.0625           # many exact ties after scaling
    x[6, 5] = np.nan; x[6, 6] = -np.nan; x[6, 7] = 1.0
    x[7, 128:] = np.float32(1e-45)                 # denormal inputs
    xt = torch.from_numpy(x).cuda()
    _check(dga, oracle, xt, dga.per_token_cast_to_fp8, oracle.quant_1x128)
    big = np.zeros((128, 128), np.float32); big[:8, :] = x[:, :128]
    big[6] = 0.0                                   # (NaN covered per token; the block amax ignores it the same way)
    _check(dga, oracle, torch.from_numpy(big).cuda(), dga.per_block_cast_to_fp8, oracle.quant_128x128)


def test_cast_feeds_the_gemm(dga, oracle):
    """End to end: bf16 activations / weights -> quantisers -> fp8 GEMM equals the oracle run on the oracle's
    quantisation of the same values."""
    m, n, k = 96, 256, 512
    g = torch.Generator(device="cuda").manual_seed(4)
    xa = torch.randn((m, k), device="cuda", generator=g).bfloat16()
    xb = (torch.randn((n, k), device="cuda", generator=g) * 0.05).bfloat16()
    qa, sfa = dga.per_token_cast_to_fp8(xa)
    qb, sfb = dga.per_block_cast_to_fp8(xb)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((qa, sfa), (qb, sfb), out, sync=True)
    a, s_a = oracle.quant_1x128(xa.float().cpu().numpy())
    b, s_b = oracle.quant_128x128(xb.float().cpu().numpy())
    want = oracle.gemm_fp8_fp8_bf16_nt(a, s_a, b, s_b)
    oracle.assert_parity(out.view(torch.int16).cpu().numpy().view(np.uint16), want, a, s_a, b, s_b)


def test_cast_empty_and_errors(dga):
    q, sf = dga.per_token_cast_to_fp8(torch.empty((0, 128), device="cuda"))
    assert q.shape == (0, 128) and sf.shape == (0, 1)
    with pytest.raises(dga.DGAError):
        dga.per_token_cast_to_fp8(torch.zeros((4, 128), device="cuda", dtype=torch.float64))


# ---- UE8M0 scales (dga_cast_to_fp8_*_ex, DGA_CAST_UE8M0): 2^ceil(log2(amax / 448)), the operand format of policy "fast_ue8m0"

@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,k", [(64, 512), (130, 1000), (17, 77), (256, 7168)])
def test_per_token_cast_ue8m0(dga, oracle, dtype, rows, k):
    g = torch.Generator(device="cuda").manual_seed(rows * 11 + k)
    x = (torch.randn((rows, k), device="cuda", generator=g) * 3.0).to(dtype)
    _check(dga, oracle, x, lambda t: dga.per_token_cast_to_fp8(t, use_ue8m0=True), lambda a: oracle.quant_1x128(a, ue8m0=True))
    _, sf = dga.per_token_cast_to_fp8(x, use_ue8m0=True)
    bits = sf.view(torch.int32)
    assert bool(((bits & 0x007FFFFF) == 0).all()) and bool((bits > 0).all()), "a scale is not a power of two"
    _, sf0 = dga.per_token_cast_to_fp8(x)
    assert bool((sf >= sf0).all()) and bool((sf < 2 * sf0).all()), "rounded UP, by less than a factor of two"


@pytest.mark.parametrize("rows,k", [(128, 128), (200, 1000), (1024, 2048)])
def test_per_block_cast_ue8m0(dga, oracle, rows, k):
    g = torch.Generator(device="cuda").manual_seed(rows * 13 + k)
    x = (torch.randn((rows, k), device="cuda", generator=g) * 0.02).to(torch.bfloat16)
    _check(dga, oracle, x, lambda t: dga.per_block_cast_to_fp8(t, use_ue8m0=True), lambda a: oracle.quant_128x128(a, ue8m0=True))
    q, sf = dga.per_block_cast_to_fp8(x, aligned_rows=True, use_ue8m0=True)        # the zero-tailed row form takes the flag too
    q2, sf2 = dga.per_block_cast_to_fp8(x, use_ue8m0=True)
    assert torch.equal(q.view(torch.uint8), q2.view(torch.uint8)) and torch.equal(sf, sf2)


def test_cast_ue8m0_exact_powers_and_edges(dga, oracle):
    """amax / 448 already a power of two stays; an all-zero block has scale 1; NaN is ignored by the amax as in the plain form."""
    x = np.zeros((4, 256), np.float32)
    x[0, 0] = 448.0 * 0.25; x[0, 1] = 1.0          # amax / 448 = 2^-2 exactly: not bumped
    x[1, 0] = np.nextafter(np.float32(448.0 * 0.25), np.float32(1e9)); x[1, 3] = -5.0     # one ulp above: the next power
    x[2, 128:] = np.linspace(-3, 3, 128)
    x[3, 7] = np.nan; x[3, 8] = 2.5
    xt = torch.from_numpy(x).cuda()
    _check(dga, oracle, xt, lambda t: dga.per_token_cast_to_fp8(t, use_ue8m0=True), lambda a: oracle.quant_1x128(a, ue8m0=True))
    _, sf = dga.per_token_cast_to_fp8(xt, use_ue8m0=True)
    s = sf.cpu().numpy()
    assert s[0, 0] == 0.25 and s[1, 0] == 0.5 and s[0, 1] == 1.0 and s[2, 0] == 1.0


def test_ue8m0_quantisers_feed_the_hardware_scale_gemm(dga, oracle):
    """bf16 values -> use_ue8m0 quantisers -> policy "fast_ue8m0" = the oracle on the oracle's ue8m0 quantisation of the same values."""
    m, n, k = 300, 520, 1152
    g = torch.Generator(device="cuda").manual_seed(3)
    xa = torch.randn((m, k), device="cuda", generator=g).to(torch.bfloat16)
    xb = (torch.randn((n, k), device="cuda", generator=g) * 0.05).to(torch.bfloat16)
    qa, sa = dga.per_token_cast_to_fp8(xa, use_ue8m0=True)
    qb, sb = dga.per_block_cast_to_fp8(xb, use_ue8m0=True)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((qa, sa), (qb, sb), out, policy="fast_ue8m0", sync=True)
    a, sfa = oracle.quant_1x128(xa.float().cpu().numpy(), ue8m0=True)
    b, sfb = oracle.quant_128x128(xb.float().cpu().numpy(), ue8m0=True)
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8)
    oracle.assert_parity(out.view(torch.int16).cpu().numpy().view(np.uint16), want, a, sfa, b, sfb)
