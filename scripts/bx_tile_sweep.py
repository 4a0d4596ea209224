"""bf16-exact policy: every build of its menu (x the tiling's split-K factor and 1) on a list of shapes, warm -- the data behind
find_bf16x_variant's mapping in dga_launch.hip (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga
from scripts.policy_perf import time_us

shapes = [(4096, 4096, 4096), (4096, 2048, 7168), (2048, 4096, 7168), (1024, 4096, 7168), (1024, 18432, 7168), (512, 4096, 7168),
          (256, 7168, 4096), (128, 4096, 7168), (128, 18432, 7168), (128, 7168, 18432), (64, 7168, 18432), (64, 18432, 7168),
          (64, 4096, 7168), (64, 24576, 1536), (64, 32768, 512), (8, 18432, 7168), (8, 7168, 18432), (2048, 2048, 2048), (768, 768, 4096)]
for (m, n, k) in shapes:
    a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128), n, k, seed=0)
    a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    base = dga.tiling(m, n, k)
    fast = time_us(lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out), 50, 100)
    auto = time_us(lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact"), 50, 100)
    ta = dga.tiling(m, n, k, policy="bf16_exact")
    line = f"{m}x{n}x{k}: fast {fast:.1f} us (tile {base.m1}x{base.n1} sk{base.splitkFactor}); bf16_exact auto {auto:.1f} ({ta.m1}x{ta.n1} sk{ta.splitkFactor}) |"
    best = (1e9, "")
    for m1, n1 in ((128, 256), (128, 128), (64, 256), (64, 128), (32, 128)):
        for sk in sorted({1, int(base.splitkFactor), 2, 3, 4, 6, 8}):
            t = dga.tiling(m, n, k)
            t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.swizzleOffset = m1, n1, sk, (4 if sk > 1 else 0), 4
            tiles = -(-m // m1) * -(-n // n1)
            if sk > 1 and (tiles * sk > 1024 or k // 128 < 2 * sk):
                continue
            try:
                us = time_us(lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t), 30, 60)
            except Exception as e:
                continue
            line += f" {m1}x{n1}/sk{sk} {us:.1f}"
            if us < best[0]:
                best = (us, f"{m1}x{n1}/sk{sk}")
    print(line + f" || best {best[1]} {best[0]:.1f}; auto / best = {auto / best[0]:.2f}", flush=True)
