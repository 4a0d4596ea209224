"""Decode rows of the 16-bit operator (catlass_dynamic_matmul, NT, bf16 / fp16) on the one-launch workgroup split-K
(csrc/gemm_b16_wsk_kernel.hpp): BIT IDENTITY with the tile kernel's two-launch split-K at factor 8 (same K slices, same per-slice
arithmetic, same combine order), and the operator's bar against the fp32 matmul of the same 16-bit values.  Reference counterparts:
the operator's device entry (/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/catlass_dynamic_matmul.cpp:16-45) and the fused
reduce of its Stream-K kernel (op_kernel/kernel/padding_streamk_matmul_kernel.h:92-107)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(dga, x, w, wsk):
    m, n = x.shape[0], w.shape[0]
    out = torch.full((m, n), float("nan"), dtype=x.dtype, device="cuda")
    old = {k: os.environ.get(k) for k in ("DGA_B16_WSK", "DGA_B16_PLAN")}
    try:
        os.environ["DGA_B16_WSK"] = "1" if wsk else "0"
        if not wsk:   # the tile kernel with K cut eight ways + the combine kernel
            os.environ["DGA_B16_PLAN"] = "16,128,8" if m <= 16 else "32,128,8"
        dga.catlass_dynamic_matmul(x, w.t(), out, sync=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return out


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("m,n,k", [
    (1, 80, 1024), (8, 512, 2048), (16, 1000, 4096 + 64), (5, 48, 64), (16, 16 * 256 + 16, 2048), (9, 16 * 700 + 5, 1024 + 64),
    (16, 16 * 1300, 1152), (3, 24, 128 * 9), (16, 4096, 7168),
    (7, 333, 1000),       # K % 64 != 0: the operands go through the padding pass first
    (17, 80, 1024), (32, 512, 2048), (24, 16 * 300 + 7, 1152), (32, 16 * 1300, 1088), (31, 4096, 7168), (20, 333, 1000),   # two row tiles
])
def test_bit_identical_to_the_two_launch_split_k(dga, dtype, m, n, k):
    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(dtype)
    w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(dtype)
    got = _run(dga, x, w, True)
    ref = _run(dga, x, w, False)
    assert torch.equal(got.view(torch.int16), ref.view(torch.int16)), f"{int((got != ref).sum())} of {got.numel()} outputs differ"
    want = x.float() @ w.float().t()
    tol = 2.0 ** -7 if dtype == torch.bfloat16 else 2.0 ** -9
    err = (got.float() - want).abs()
    assert bool((err <= tol * want.abs() + tol * (x.float().abs() @ w.float().abs().t()) * 2.0 ** -6).all())


def test_rows_beyond_thirty_two_keep_the_tile_kernels(dga):
    x = torch.randn((33, 2048), device="cuda").to(torch.bfloat16)
    w = torch.randn((256, 2048), device="cuda").to(torch.bfloat16)
    a = _run(dga, x, w, True)      # the request falls through (DGA_E_TILING) to the planned tile kernel
    os.environ.pop("DGA_B16_PLAN", None)
    out = torch.empty_like(a)
    dga.catlass_dynamic_matmul(x, w.t(), out, sync=True)
    assert torch.equal(a.view(torch.int16), out.view(torch.int16))
