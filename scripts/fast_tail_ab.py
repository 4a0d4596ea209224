"""Fast policy: the selector's pick against the 256 x 256 tile with its last partial round in quarter tiles (kernelSerial 5) on rasters whose
tail lies between a quarter and a half of the CUs -- the launcher takes those, the selector (apply_tail_split) stops at a quarter."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

SHAPES = [(2816, 8192, 7168), (5632, 4096, 7168), (4864, 5120, 4096), (3072, 7680, 7168), (2560, 9216, 4096), (6144, 4096, 4096),
          (1024, 18432, 7168), (5120, 5120, 5120), (4352, 4096, 7168), (4608, 4096, 7168), (5120, 4096, 7168)]
for (m, n, k) in SHAPES:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=3)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    pick = dga.tiling(m, n, k, policy="fast")
    f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=pick)
    f0(); torch.cuda.synchronize()
    us0 = min(bench._prewarmed_us(f0, 30, 100.0) for _ in range(2))
    tiles = ((m + 255) // 256) * ((n + 255) // 256)
    tail = tiles % 256
    res = []
    for tag in (2, 6):
        t = dga.tiling(4096, 4096, 4096, policy="fast")   # a 256 x 256 continuous tiling to start from
        t2 = dga.tiling(m, n, k, policy="fast")
        for f in ("m1", "n1", "wavesM", "wavesN", "stages"):
            setattr(t2, f, getattr(t, f))
        t2.dispatchPolicyTag, t2.kernelSerial, t2.splitkFactor = tag, 5, 1
        t2.blockDim = tiles - tail + 4 * tail
        if dga.tiling_check(t2) != 0:
            continue
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t2)
        fn(); torch.cuda.synchronize()
        res.append((min(bench._prewarmed_us(fn, 30, 100.0) for _ in range(2)), tag))
    res.sort()
    print(f"{m:>5}x{n:>6}x{k:>6} 256x256 tiles {tiles:>4} rounds {tiles / 256:5.2f} tail {tail:>3} | pick {pick.m1}x{pick.n1} ks{pick.kernelSerial} p{pick.dispatchPolicyTag} s{pick.splitkFactor}: {us0:8.2f} us | "
          + "  ".join(f"256x256 ks5 p{tag}: {us:8.2f} ({us / us0:.3f})" for us, tag in res), flush=True)
    del a, b, o
