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
