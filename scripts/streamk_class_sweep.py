"""Where does the one-launch Stream-K (kernelSerial 7) beat the selector's pick?  Few 256 x 256 tiles with a deep K: the class the
A/B of scripts/streamk_ab.py pointed at.  Cold-ish protocol: operand sets rotated so that B (the large operand) does not sit in the
Infinity Cache from one call to the next."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
print(f"{'shape':>22} {'tiles':>5} {'parts':>5} | {'pick':>24} {'us':>8} | {'stream-K':>9} {'ratio':>6}")
for m in (256, 512, 768, 1024):
    for n in (4096, 7168):
        for k in (4096, 8192, 16384, 18432, 32768):
            tiles = (m // 256) * (n // 256)
            if tiles > 128:
                continue
            a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=1)
            o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
            t0 = dga.tiling(m, n, k)
            tk = dga.tiling(m, n, k)
            tk.m1, tk.n1, tk.wavesM, tk.wavesN, tk.stages, tk.dispatchPolicyTag, tk.kernelSerial, tk.splitkFactor = 256, 256, 4, 2, 2, 2, 7, 1
            f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="fast")
            fk = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=tk)
            f0(); fk(); torch.cuda.synchronize()
            us0 = min(bench._prewarmed_us(f0, iters, 60.0) for _ in range(3))
            usk = min(bench._prewarmed_us(fk, iters, 60.0) for _ in range(3))
            sp = 1
            while sp * 2 * tiles <= 256 and sp * 2 <= 16 and (k // 128) // (sp * 2) >= 2:
                sp *= 2
            pick = f"{t0.m1}x{t0.n1} ks{t0.kernelSerial} p{t0.dispatchPolicyTag} s{t0.splitkFactor}"
            print(f"{m:>6}x{n:>6}x{k:>6} {tiles:>5} {sp:>5} | {pick:>24} {us0:8.2f} | {usk:9.2f} {usk / us0:6.3f}", flush=True)
            del a, b, o
