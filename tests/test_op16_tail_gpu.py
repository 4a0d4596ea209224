"""The last partial round of a 256x256 / 128x256 raster in sub-tiles (128 / 64 / 32 rows x 128 columns) (16-bit paths; csrc/dga_b16.hip `go_tail`,
gemm_b16_kernel.hpp `tail_sub`): the purpose of the reference's Stream-K handler (wave quantisation,
/root/reference/aclnn_catlass_dynamic_matmul/op_host/op_tiling/select_kernel.cpp:303-331) without partial sums -- every output
still comes from one accumulation in k order, so the bytes are those of the single launch."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _with_plan(plan, fn):
    old = os.environ.get("DGA_B16_PLAN")
    try:
        os.environ["DGA_B16_PLAN"] = plan
        return fn()
    finally:
        if old is None:
            os.environ.pop("DGA_B16_PLAN", None)
        else:
            os.environ["DGA_B16_PLAN"] = old


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile,sub", [("256,256", 128), ("256,256", 64), ("256,256", 32), ("128,256", 64), ("128,256", 32)])
@pytest.mark.parametrize("m,n,k", [(2100, 8000, 512), (4096, 4608, 256), (1100, 16500, 320), (2048, 9000, 200), (2101, 8001, 130), (3511, 6151, 72)])
def test_operator_tail_matches_the_single_launch(dga, dtype, tile, sub, m, n, k):
    if tile == "128,256" and m * n < 257 * 128 * 256:
        m = 2 * m
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(dtype)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(dtype)

    def run():
        out = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        dga.catlass_dynamic_matmul(x, w.t(), out, sync=True)
        return out
    tail = _with_plan(f"{tile},1,{sub}", run)
    whole = _with_plan(tile + ",1,0", run)
    assert torch.isfinite(tail.float()).all()
    assert torch.equal(tail.view(torch.int16), whole.view(torch.int16)), f"{int((tail != whole).sum())} outputs differ"
    want = x.float() @ w.float().t()
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert bool(((tail.float() - want).abs() <= tol * want.abs() + 2.0 ** -12 * (x.float().abs() @ w.float().abs().t())).all())


@pytest.mark.parametrize("tile,sub", [("256,256", 128), ("256,256", 64), ("128,256", 64), ("128,256", 32)])
@pytest.mark.parametrize("m,n,k", [(2100, 8000, 512), (2304, 8192, 192), (2048, 9001, 200)])
def test_run_mmad_rtc_tail_matches_the_single_launch(dga, tile, sub, m, n, k):
    if tile == "128,256":
        m = 2 * m
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn((1, m, k), device="cuda", generator=g) * 0.5).to(torch.float16)
    y = (torch.randn((1, k, n), device="cuda", generator=g) * 0.5).to(torch.float16)

    def run():
        z = torch.full((1, m, n), float("nan"), dtype=torch.float32, device="cuda")
        dga.run_mmad_rtc(x, y, z)
        return z
    tail = _with_plan(f"{tile},1,{sub}", run)
    whole = _with_plan(tile + ",1,0", run)
    assert torch.equal(tail.view(torch.int32), whole.view(torch.int32))
    want = x[0].float() @ y[0].float()
    assert bool(((tail[0] - want).abs() <= 2.0 ** -16 * (x[0].float().abs() @ y[0].float().abs())).all())
