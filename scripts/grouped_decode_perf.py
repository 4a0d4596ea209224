"""Grouped masked-M GEMM with few rows per expert (decode-time MoE): is the weight stream the only cost? (development aid)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import parallel
from widen_perf import timeit
g = torch.Generator(device="cuda").manual_seed(0)
G, MM, N, K = 256, 128, 2048, 7168
a = parallel._rand_fp8((G, MM, K), g, "cuda"); b = parallel._rand_fp8((G, N, K), g, "cuda")
sfa = torch.rand((G, MM, K // 128), device="cuda") + 0.5; sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
out = torch.empty((G, MM, N), dtype=torch.bfloat16, device="cuda")
for exp_m, hi in ((128, 128), (64, 128), (32, 64), (16, 32), (8, 16), (4, 8)):
    mask = torch.randint(0, hi + 1, (G,), device="cuda", generator=g).int()
    t = dga.select_kernel(MM, N, K, groups=G, expected_m=exp_m)
    us = min(timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, mask, exp_m, tiling_=t), iters=10, warm=3) for _ in range(3))
    act = int((mask > 0).sum()); rows = int(mask.sum())
    byt = act * N * K + rows * (K + 224 + 2 * N)
    print(f"expected_m {exp_m} (masks 0..{hi}, {rows} rows, {act} active experts): tile {t.m1}x{t.n1} st{t.stages}: {us:.0f} us  {byt/us/1e3:.0f} GB/s", flush=True)
