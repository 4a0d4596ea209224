"""Decode / short-M grid under the bf16-exact policy: the selector's pick against the one-launch split-K of the 64 x 128 tile (build
DGA_BUILD_BX_DECODE) at the rule's split count S = min(8, CUs / tiles, KB / 4) and at half of it; graph replay, one process."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import _lib

Ms = (40, 64, 96, 128, 192, 256)
NKs = ((2112, 7168), (4096, 7168), (7168, 2048), (7168, 4096), (4096, 4096), (4096, 2048), (24576, 1536), (32768, 512), (7168, 18432), (18432, 7168),
       (1536, 7168), (3072, 1536), (16384, 7168), (4096, 14336), (2048, 7168), (5120, 5120), (8192, 1024), (576, 7168), (12288, 5120))
g = torch.Generator(device="cuda").manual_seed(1)
geo = []
for (n, k) in NKs:
    for m in Ms:
        kb = -(-k // 128)
        tiles = -(-m // 64) * -(-n // 128)
        if tiles > 256:
            continue
        a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=g)
        b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=g)
        sfa = torch.rand((m, kb), device="cuda", generator=g) + 0.5
        sfb = torch.rand((-(-n // 128), kb), device="cuda", generator=g) + 0.5
        out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        import os
        os.environ["DGA_NO_DSK_PICK"] = "1"
        t0 = dga.tiling(m, n, k, policy="bf16_exact")
        if t0.build == _lib.BUILD_BX_DECODE:
            continue
        f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t0)
        f0(); torch.cuda.synchronize()
        us0 = bench._graph_us(f0, 20)
        smax = max(1, min(8, 256 // tiles, kb // 4))
        row = {}
        for s in sorted({smax, max(1, smax // 2), max(1, (smax * 3) // 4)}):
            if kb < 4 * s:
                continue
            t = dga.tiling(m, n, k, policy="bf16_exact")
            t.m1, t.n1, t.kernelSerial, t.build, t.splitkFactor, t.stages = 64, 128, 6, _lib.BUILD_BX_DECODE, s, 0
            f = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
            f(); torch.cuda.synchronize()
            row[s] = bench._graph_us(f, 20)
        sb = min(row, key=row.get)
        print(f"{m}x{n}x{k} tiles {tiles} kb {kb} pick {t0.m1}x{t0.n1} ser{t0.kernelSerial} s{t0.splitkFactor}: {us0:.2f} | dsk " +
              " ".join(f"s{s}:{v:.2f}" for s, v in row.items()) + f" | best/pick {row[sb] / us0:.3f} smax/pick {row[smax] / us0:.3f} nb {kb / (2 * smax):.1f}", flush=True)
