"""Two-launch split-K picks of the bf16-exact policy (and the fast one) on decode / mid-M rows, graph replay: run once as it is and once
with DGA_OUT_NT=0 (slab rows stored plain instead of write-through) in the same gpurun call to see what the slab store policy is worth."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["DGA_NO_DSK_PICK"] = "1"
import torch
import bench
import deepgemm_ascend_amd as dga

shapes = [(64, 4096, 7168), (128, 4096, 7168), (64, 7168, 18432), (64, 18432, 7168), (128, 7168, 18432), (64, 2112, 7168), (32, 4096, 7168),
          (256, 4096, 7168), (512, 4096, 7168), (64, 7168, 2048)]
g = torch.Generator(device="cuda").manual_seed(1)
for pol in ("bf16_exact", "fast"):
    for (m, n, k) in shapes:
        kb = -(-k // 128)
        a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=g)
        b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=g)
        sfa = torch.rand((m, kb), device="cuda", generator=g) + 0.5
        sfb = torch.rand((-(-n // 128), kb), device="cuda", generator=g) + 0.5
        out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        t = dga.tiling(m, n, k, policy=pol)
        f = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy=pol, tiling_=t)
        f(); torch.cuda.synchronize()
        us = min(bench._graph_us(f, 20) for _ in range(3))
        print(f"{pol} {m}x{n}x{k} {t.m1}x{t.n1} ser{t.kernelSerial} s{t.splitkFactor}: {us:.2f}", flush=True)
