"""One-launch Stream-K (kernelSerial 7, 256 x 256 tile) against the selector's pick on rasters that do not fill whole rounds of
the chip: time per call, parity against the strict kernel, and the same under the hardware-scale flag (power-of-two scales).
  python scripts/streamk_ab.py [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
SHAPES = [(1024, 18432, 7168), (2048, 5120, 13824), (1024, 4096, 7168), (2048, 4096, 7168), (1280, 4096, 7168), (3072, 4096, 7168),
          (4352, 4096, 4096), (4096, 4608, 7168), (2304, 4608, 7168), (4096, 4096, 4096), (512, 7168, 18432), (1024, 7168, 18432),
          (2048, 18432, 7168), (8192, 8192, 8192)]


def key(t):
    v = t.view(torch.int16).to(torch.int32)
    mag = v & 0x7FFF
    return torch.where(v < 0, -mag, mag)


print(f"{'shape':>22} {'tiles':>6} {'rounds':>6} | {'pick':>22} {'us':>8} | {'stream-K us':>11} {'ratio':>6} {'max_ulp':>7} {'>2ulp':>9} | {'ue8m0 pick':>10} {'ue8m0 sk':>9}")
for (m, n, k) in SHAPES:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=2, ue8m0=True)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda"); o_s = torch.empty_like(o); o_k = torch.empty_like(o)
    t0 = dga.tiling(m, n, k)
    tk = dga.tiling(m, n, k)
    tk.m1, tk.n1, tk.wavesM, tk.wavesN, tk.stages, tk.dispatchPolicyTag, tk.kernelSerial, tk.splitkFactor = 256, 256, 4, 2, 2, 2, 7, 1
    f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="fast")
    fk = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_k, tiling_=tk)
    f0(); fk()
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_s, strict=True, sync=True)
    ul = (key(o_k) - key(o_s)).abs()
    us0 = min(bench._prewarmed_us(f0, iters, 100.0) for _ in range(2))
    usk = min(bench._prewarmed_us(fk, iters, 100.0) for _ in range(2))
    tku = dga.tiling(m, n, k); 
    tku.m1, tku.n1, tku.wavesM, tku.wavesN, tku.stages, tku.dispatchPolicyTag, tku.kernelSerial, tku.splitkFactor = 256, 256, 4, 2, 2, 2 | 16, 7, 1
    f0u = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="fast_ue8m0")
    fku = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_k, tiling_=tku)
    f0u(); fku(); torch.cuda.synchronize()
    us0u = min(bench._prewarmed_us(f0u, iters, 100.0) for _ in range(2))
    usku = min(bench._prewarmed_us(fku, iters, 100.0) for _ in range(2))
    tiles = (m // 256) * (n // 256)
    pick = f"{t0.m1}x{t0.n1} ks{t0.kernelSerial} p{t0.dispatchPolicyTag} s{t0.splitkFactor}"
    print(f"{m:>6}x{n:>6}x{k:>6} {tiles:>6} {tiles / 256:6.3f} | {pick:>22} {us0:8.2f} | {usk:11.2f} {usk / us0:6.3f} {int(ul.max()):>7} {float((ul > 2).double().mean()):9.2e} | "
          f"{us0u:10.2f} {usku:9.2f}", flush=True)
    del a, b, o, o_s, o_k
