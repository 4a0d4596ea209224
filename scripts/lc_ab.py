"""128x256 three-stage tile: 8 computing waves (2x4) against 4 computing + 4 loader waves (dispatchPolicyTag 4), on the
dense shapes whose tiling names that tile (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit
import os
if os.environ.get("LC_AB_SMALL"):
    for (m, n, k) in [(2048, 2048, 7168), (1024, 4096, 7168), (1024, 2048, 7168), (512, 4096, 7168), (256, 8192, 7168), (64, 18432, 7168), (64, 24576, 1536)]:
        a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128), n, k, seed=0)
        a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        t0 = dga.select_kernel(m, n, k)
        line = f"{m}x{n}x{k}: heuristic {t0.m1}x{t0.n1} w{t0.wavesM}x{t0.wavesN} st{t0.stages} p{t0.dispatchPolicyTag} sk{t0.splitkFactor}"
        for name, over in {"heuristic": None, "128x128 st3": (128, 128, 2, 2, 3, 0), "128x128 st3 +loaders": (128, 128, 2, 2, 3, 4),
                           "64x256 st3": (64, 256, 1, 4, 3, 0), "64x256 st3 +loaders": (64, 256, 1, 4, 3, 4)}.items():
            t = dga.select_kernel(m, n, k)
            if over:
                if over[0] >= 2 * m: continue
                t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = over
                t.kernelSerial, t.splitkFactor = 0, 1
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
            line += f" | {name} {min(timeit(fn, iters=60, warm=100) for _ in range(3)):.1f}"
        print(line, flush=True)
    sys.exit(0)
shapes = [(4096, 2048, 7168), (2048, 4096, 7168), (4096, 4096, 2048), (2048, 2048, 7168), (8192, 2048, 7168), (4096, 4096, 7168),
          (1024, 8192, 7168), (3072, 2048, 7168), (4096, 1536, 7168), (2048, 7168, 2048), (6144, 2048, 4096), (1024, 4096, 7168),
          (512, 4096, 7168), (8192, 4096, 4096)]
for (m, n, k) in shapes:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t0 = dga.select_kernel(m, n, k)
    res = {}
    for name, over in {"heuristic": None, "2x4": (128, 256, 2, 4, 3, 0), "2x2+loaders": (128, 256, 2, 2, 3, 4)}.items():
        t = dga.select_kernel(m, n, k)
        if over:
            t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag = over
            t.kernelSerial, t.splitkFactor = 0, 1
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
        res[name] = min(timeit(fn, iters=60, warm=100) for _ in range(3))
    tiles = -(-m // 128) * -(-n // 256)
    print(f"{m}x{n}x{k} tiles128x256 {tiles}: heuristic {t0.m1}x{t0.n1} w{t0.wavesM}x{t0.wavesN} st{t0.stages} p{t0.dispatchPolicyTag} sk{t0.splitkFactor} {res['heuristic']:.1f} us | 2x4 {res['2x4']:.1f} | 2x2+loaders {res['2x2+loaders']:.1f}", flush=True)
