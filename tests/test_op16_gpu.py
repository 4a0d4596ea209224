"""GPU parity of the aclnn operator in its own dtypes (CatlassDynamicMatmul: fp16/bf16 in, same dtype out, NT layout),
judged as the reference's own harness judges it: fp32 numpy golden of the up-cast inputs, np.isclose with the dtype's
rtol, mismatch fraction <= 1e-4 (scripts/verify.py:14-35; custom_catlass golden CompareData)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
RTOL = {torch.float16: 2.0 ** -10, torch.bfloat16: 2.0 ** -7}   # one rounding of the output type


def _case(dtype, m, n, k, seed):
    g = torch.Generator().manual_seed(seed)
    a = (torch.randn((m, k), generator=g) * 0.5).to(dtype)
    b = (torch.randn((n, k), generator=g) * 0.5).to(dtype)
    golden = a.float().numpy() @ b.float().numpy().T
    return a, b, golden


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("m,n,k", [(128, 128, 128), (1024, 1024, 1024), (2048, 2048, 512),   # in place, 128^2 / 256^2 tiles
                                   (4096, 2048, 192), (4000, 2000, 72),                        # 128x256, 8 waves, three stages
                                   (300, 520, 200), (257, 129, 100),                          # padded copies, ragged edges
                                   (8, 1024, 4096), (40, 640, 2048), (128, 512, 8192),        # short tiles + split-K
                                   (1, 16, 64), (5, 7, 3)])
def test_catlass_dynamic_matmul(dga, oracle, dtype, m, n, k):
    a, b, golden = _case(dtype, m, n, k, seed=m + 3 * n + 7 * k)
    out = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
    dga.catlass_dynamic_matmul(a.cuda(), b.cuda().t(), out, sync=True)
    ok, ratio = oracle.verify_isclose(out.float().cpu().numpy(), golden, rtol=RTOL[dtype], atol=1e-3 * np.sqrt(k))
    assert ok, ratio


def test_matches_the_fp32_output_path_after_one_rounding(dga):
    """Same accumulation as run_mmad_rtc (fp32 out): the 16-bit result is its RNE rounding, bit for bit, on an aligned
    shape where both take the same tile and no split-K."""
    m, n, k = 512, 1024, 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn((m, k), device="cuda", generator=g).bfloat16(); b = torch.randn((n, k), device="cuda", generator=g).bfloat16()
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    dga.catlass_dynamic_matmul(a, b.t(), out, sync=True)
    z = torch.empty((1, m, n), dtype=torch.float32, device="cuda")
    dga.run_mmad_rtc(a[None], b.t().contiguous()[None], z)
    assert torch.equal(out, z[0].bfloat16())


def test_without_workspace_and_errors(dga, oracle):
    import ctypes
    from deepgemm_ascend_amd import _lib
    a, b, golden = _case(torch.bfloat16, 70, 90, 72, seed=2)        # K % 64 != 0 and no workspace: element-wise kernel
    out = torch.zeros((70, 90), dtype=torch.bfloat16, device="cuda")
    ac, bc = a.cuda(), b.cuda()
    rc = _lib.lib().dga_catlass_dynamic_matmul(ac.data_ptr(), bc.data_ptr(), out.data_ptr(), 70, 90, 72, _lib.DT_BF16, None, 0, None)
    torch.cuda.synchronize()
    assert rc == 0
    assert oracle.verify_isclose(out.float().cpu().numpy(), golden, rtol=2.0 ** -7, atol=1e-2)[0]
    with pytest.raises(dga.DGAError):   # row-major mat2 is not the operator's layout
        dga.catlass_dynamic_matmul(ac, torch.zeros((72, 90), dtype=torch.bfloat16, device="cuda"), out)
    with pytest.raises(dga.DGAError):   # dtype mismatch (InferDataType: inputs must match)
        dga.catlass_dynamic_matmul(ac, bc.half().t(), out)
    empty = torch.empty((0, 90), dtype=torch.bfloat16, device="cuda")
    dga.catlass_dynamic_matmul(torch.empty((0, 72), dtype=torch.bfloat16, device="cuda"), bc.t(), empty)


@pytest.mark.parametrize("m,n,k", [(8, 512, 1000), (8, 2048, 2048), (24, 640, 4096), (64, 4096, 1024), (300, 8000, 128), (2100, 8000, 128)])
def test_without_workspace_on_the_round_4_plans(dga, m, n, k):
    """No workspace (a legal call of the C entry): decode rows with odd K (no room for the padded copies: element-wise kernel) and with
    whole k steps (the one-launch workgroup split-K needs none), a plan that would split K (runs unsplit), a raster with a sub-tile
    tail (two launches, no slab): every one against the fp32 matmul of the same bf16 values."""
    from deepgemm_ascend_amd import _lib
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16)
    out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device="cuda")
    rc = _lib.lib().dga_catlass_dynamic_matmul(x.data_ptr(), w.data_ptr(), out.data_ptr(), m, n, k, _lib.DT_BF16, None, 0, None)
    torch.cuda.synchronize()
    assert rc == 0
    want = x.float() @ w.float().t()
    assert bool(((out.float() - want).abs() <= 2.0 ** -7 * want.abs() + 2.0 ** -12 * (x.float().abs() @ w.float().abs().t())).all())
