"""In-process A/B of dense-kernel schedules (same device, interleaved rounds)."""
import sys, os
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 2048, 7168), (1024, 4096, 7168)]
for (m, n, k) in shapes:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    ref = torch.empty_like(out)
    variants = {}
    for name, (tile, pp) in {"256x256": ((256, 256), 0), "256x256-pp": ((256, 256), 1), "256x256-cont": ((256, 256), 2),
                             "128x256": ((128, 256), 0), "128x256-cont": ((128, 256), 2),
                             "128x128": ((128, 128), 0), "128x128-cont": ((128, 128), 2), "auto": (None, -1)}.items():
        t = dga.tiling(m, n, k)
        if tile:
            t.m1, t.n1 = tile; t.stages = 2; t.wavesM = t.wavesN = 0; t.splitkFactor = 1; t.kernelSerial = 0
        if pp >= 0: t.dispatchPolicyTag = pp
        variants[name] = t
    res = {k_: [] for k_ in variants}
    t0 = variants["256x256"]
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), ref, tiling_=t0)
    for name, t in variants.items():
        out.zero_()
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t, sync=True)
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
            d = (out.float() - ref.float()).abs().max().item()
            print(f"  !! {name} differs from 256x256 baseline: max abs {d}")
    for rnd in range(5):
        for name, t in variants.items():
            for _ in range(3): dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 50)
    for name, v in res.items():
        v = sorted(v); med = v[len(v) // 2]
        print(f"{m}x{n}x{k} {name:12s} tile {variants[name].m1}x{variants[name].n1}: median {med:.1f} us  min {v[0]:.1f} us  {2*m*n*k/med/1e6:.0f} TF (best {2*m*n*k/v[0]/1e6:.0f})", flush=True)
