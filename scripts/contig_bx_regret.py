"""Contiguous-grouped layout under the in-contract (bf16-exact) policy: the tile the policy's tiling keeps (the fast tiling's, mapped onto
this policy's menu) against every tile of that menu no taller than the segment alignment."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

for (groups, per, n, k) in [(8, 1024, 4096, 7168), (32, 256, 4096, 7168), (64, 128, 7168, 2048), (8, 512, 7168, 2048), (16, 384, 2048, 7168), (256, 128, 7168, 2048)]:
    gen = torch.Generator(device="cuda").manual_seed(7)
    msum = groups * per
    a = torch.randint(0, 120, (msum, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 120, (groups, n, k), dtype=torch.uint8, device="cuda", generator=gen)
    sfa = torch.rand((msum, k // 128), device="cuda") + 0.5
    sfb = torch.rand((groups, n // 128, k // 128), device="cuda") + 0.5
    idx = torch.arange(groups, device="cuda", dtype=torch.int32).repeat_interleave(per).contiguous()
    out = torch.empty((msum, n), dtype=torch.bfloat16, device="cuda")
    pick = dga.tiling(msum, n, k, groups=groups, contiguous=True, policy="bf16_exact")
    f0 = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=pick)
    f0(); torch.cuda.synchronize()
    ref = out.clone()
    us0 = min(bench._prewarmed_us(f0, 20, 100.0) for _ in range(2))
    rows = []
    for (bm, bn) in [(128, 256), (128, 128), (64, 256), (64, 128), (32, 128)]:
        for st in (3, 7, 8):
            if st == 7 and (bm, bn) != (128, 256):
                continue
            t = dga.tiling(msum, n, k, groups=groups, contiguous=True, policy="bf16_exact")
            t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.splitkFactor, t.kernelSerial = bm, bn, st, 0, 0, 1, 0
            if dga.tiling_check(t) != 0:
                continue
            fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t)
            try:
                fn(); torch.cuda.synchronize()
            except Exception:
                continue
            same = bool(torch.equal(out.view(torch.int16), ref.view(torch.int16)))
            us = min(bench._prewarmed_us(fn, 20, 100.0) for _ in range(2))
            rows.append((us, bm, bn, st, same))
    rows.sort()
    print(f"{groups} groups x {per} rows, N {n} K {k}: pick {pick.m1}x{pick.n1} st{pick.stages} p{pick.dispatchPolicyTag}: {us0:8.1f} us ({2.0 * msum * n * k / us0 / 1e6:6.0f} TF) | "
          + "  ".join(f"{r[1]}x{r[2]}st{r[3]}:{r[0]:.1f}{'' if r[4] else '!'}" for r in rows[:6]) + f" | regret {100 * (us0 / rows[0][0] - 1):.1f} %", flush=True)
    del a, b, out, ref
