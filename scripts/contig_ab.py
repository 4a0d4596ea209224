"""Contiguous-grouped layout: 128-row tiles vs 256-row two-pass tiles (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from widen_perf import timeit, rand_fp8

g = torch.Generator(device="cuda").manual_seed(0)
for groups, per, n, k in ((8, 1024, 4096, 7168), (8, 1024, 7168, 2048), (16, 512, 4096, 7168), (4, 8192, 4096, 7168), (8, 1000, 4096, 7168), (32, 256, 4096, 7168)):
    seg = -(-per // 128) * 128
    msum = groups * seg
    a = rand_fp8((msum, k), g); b = rand_fp8((groups, n, k), g)
    sfa = torch.rand((msum, k // 128), device="cuda") + 0.5
    sfb = torch.rand((groups, n // 128, k // 128), device="cuda") + 0.5
    idx = torch.full((groups, seg), -1, dtype=torch.int32, device="cuda")
    for i in range(groups):
        idx[i, :per] = i
    idx = idx.reshape(-1).contiguous()
    outs = {}
    for name, (bm, bn, st, pol, wm, wn) in {"128x256": (128, 256, 3, 0, 2, 2), "128x256 8w": (128, 256, 3, 0, 2, 4), "256x256": (256, 256, 2, 2, 0, 0)}.items():
        t = dga.tiling(msum, n, k, groups=groups, contiguous=True)
        t.m1, t.n1, t.stages, t.dispatchPolicyTag, t.wavesM, t.wavesN = bm, bn, st, pol, wm, wn
        out = torch.zeros((msum, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t)
        timeit(fn, iters=10, warm=5)
        us = min(timeit(fn, iters=20, warm=3) for _ in range(3))
        outs[name] = out
        print(f"G={groups} x {per} rows N={n} K={k} tile {name}: {us:.1f} us  {2.0 * groups * per * n * k / us / 1e6:.0f} TFLOP/s (valid rows)", flush=True)
    print("   equal:", torch.equal(outs["128x256"].view(torch.int16), outs["256x256"].view(torch.int16)), flush=True)
