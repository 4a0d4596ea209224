"""Decode rows under the bf16-exact policy: the one-launch split-K of the 64 x 128 tile (kernelSerial 6, build DGA_BUILD_BX_DECODE;
gemm_fp8_bf16x_dsk_kernel.hpp) against the selector's pick before it (two-launch split-K / tile kernel), one process, graph replay.
Also checks the new build against the selector's pick: every output within one bf16 ULP (the partial sums are rounded in another order)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import bench
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import _lib

shapes = [(64, 4096, 7168), (128, 4096, 7168), (48, 4096, 7168), (64, 7168, 18432), (64, 18432, 7168), (64, 7168, 16384), (128, 7168, 18432),
          (64, 24576, 1536), (33, 2112, 7168), (64, 4096, 4096), (64, 7168, 2048), (128, 2112, 7168), (64, 4000, 7100)]
if len(sys.argv) > 1:
    shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]]
g = torch.Generator(device="cuda").manual_seed(1)
for (m, n, k) in shapes:
    kb = -(-k // 128)
    a = torch.randint(0, 127, (m, k), dtype=torch.uint8, device="cuda", generator=g)
    a |= torch.randint(0, 2, (m, k), dtype=torch.uint8, device="cuda", generator=g) << 7
    b = torch.randint(0, 127, (n, k), dtype=torch.uint8, device="cuda", generator=g)
    b |= torch.randint(0, 2, (n, k), dtype=torch.uint8, device="cuda", generator=g) << 7
    sfa = torch.rand((m, kb), device="cuda", generator=g) + 0.5
    sfb = torch.rand((-(-n // 128), kb), device="cuda", generator=g) + 0.5
    t0 = dga.tiling(m, n, k, policy="bf16_exact")
    ref = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), ref, policy="bf16_exact", tiling_=t0)
    f0(); torch.cuda.synchronize()
    row = {"pick": f"{t0.m1}x{t0.n1} ser{t0.kernelSerial} s{t0.splitkFactor} b{t0.build}", "us_pick": round(bench._graph_us(f0, 20), 2)}
    tiles = -(-m // 64) * -(-n // 128)
    best = None
    for s in (1, 2, 3, 4, 6, 7, 8):
        if tiles * s > 256 or kb < 4 * s:
            continue
        t = dga.tiling(m, n, k, policy="bf16_exact")
        t.m1, t.n1, t.kernelSerial, t.build, t.splitkFactor, t.stages = 64, 128, 6, _lib.BUILD_BX_DECODE, s, 0
        out = torch.full((m, n), -3.0, dtype=torch.bfloat16, device="cuda")
        f = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
        f(); torch.cuda.synchronize()
        x, y = out.view(torch.int16).cpu().numpy().astype(np.int32), ref.view(torch.int16).cpu().numpy().astype(np.int32)
        bad = int((np.abs(x - y) > 1).sum()); diff = int((x != y).sum())
        out2 = torch.full((m, n), -3.0, dtype=torch.bfloat16, device="cuda")
        f2 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out2, policy="bf16_exact", tiling_=t)
        f2(); torch.cuda.synchronize()
        same = bool(torch.equal(out.view(torch.int16), out2.view(torch.int16)))
        us = bench._graph_us(f, 20)
        row[f"s{s}"] = f"{us:.2f}" + ("" if bad == 0 and same else f" BAD({bad},{same})") + f" d{diff}"
    print(f"{m}x{n}x{k}", row, flush=True)
