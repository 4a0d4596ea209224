"""Every (tile, split-K) plan the 16-bit paths can be given computes the same product (csrc/dga_b16.hip `b16_plan`, forced through
$DGA_B16_PLAN): the operator (catlass_dynamic_matmul, NT, 16-bit out; reference device entry
/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/catlass_dynamic_matmul.cpp:16-45) and run_mmad_rtc (y [K,N], f32 out;
/root/reference/deep_gemm_ascend/framework/csrc/jit_kernels/impls/gemm.hpp:68-111) against the fp32 matmul of the same 16-bit values.
(The 256x256 tile's continuous loop once ignored its K slice: a plan with that tile and split-K summed the whole K once per slice.)"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = ["16,128", "32,128", "64,128", "128,128", "128,256", "256,256", "128,128;w8"]   # ;w8 = the 8-wave build of the 128 x 128 tile
SPLITS = [1, 2, 3, 5]


def _with_plan(plan, fn, deep=None):
    """deep: $DGA_B16_DEEP -- the four-stage build of a tile of at most 64 rows forced on ("1") or off ("0")."""
    old = {k: os.environ.get(k) for k in ("DGA_B16_PLAN", "DGA_B16_DEEP")}
    try:
        os.environ["DGA_B16_PLAN"] = plan
        if deep is not None:
            os.environ["DGA_B16_DEEP"] = deep
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("deep", ["0", "1"])
def test_operator_every_plan(dga, dtype, tile, split, deep):
    m, n, k = 300, 520, 1344     # ragged in every dimension of every tile; 21 k steps: uneven slices
    g = torch.Generator(device="cuda").manual_seed(split)
    x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(dtype)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(dtype)

    def run():
        out = torch.full((m, n), float("nan"), dtype=dtype, device="cuda")
        dga.catlass_dynamic_matmul(x, w.t(), out, sync=True)
        return out
    got = _with_plan(f"{tile.split(';')[0]},{split}" + (",0,1" if ";" in tile else ""), run, deep)
    want = x.float() @ w.float().t()
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -10
    assert bool(((got.float() - want).abs() <= tol * want.abs() + 2.0 ** -12 * (x.float().abs() @ w.float().abs().t())).all())


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("split", SPLITS)
@pytest.mark.parametrize("n", [520, 523])       # y read where it lies / through the transposing pre-pass
@pytest.mark.parametrize("deep", ["0", "1"])
def test_run_mmad_rtc_every_plan(dga, tile, split, n, deep):
    m, k = 300, 1344
    g = torch.Generator(device="cuda").manual_seed(split + n)
    x = (torch.randn((2, m, k), device="cuda", generator=g) * 0.5).to(torch.float16)
    y = (torch.randn((2, k, n), device="cuda", generator=g) * 0.5).to(torch.float16)
    z = torch.full((2, m, n), float("nan"), dtype=torch.float32, device="cuda")
    _with_plan(f"{tile.split(';')[0]},{split}" + (",0,1" if ";" in tile else ""), lambda: dga.run_mmad_rtc(x, y, z), deep)
    for b in range(2):
        want = x[b].float() @ y[b].float()
        assert bool(((z[b] - want).abs() <= 2.0 ** -16 * (x[b].float().abs() @ y[b].float().abs())).all())


def test_auto_plan_on_the_shape_the_fuzz_found(dga):
    m, n, k = 1000, 4100, 4096     # the planner picks the 256x256 tile with split-K here
    g = torch.Generator(device="cuda").manual_seed(0)
    x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.float16)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.float16)
    out = torch.full((m, n), float("nan"), dtype=torch.float16, device="cuda")
    dga.catlass_dynamic_matmul(x, w.t(), out, sync=True)
    want = x.float() @ w.float().t()
    assert bool(((out.float() - want).abs() <= 2.0 ** -10 * want.abs() + 2.0 ** -12 * (x.float().abs() @ w.float().abs().t())).all())
