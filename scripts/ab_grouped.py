"""In-process A/B of kernel variants on the grouped masked-M workload (BASELINE configs[3])."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
G, MMAX, N, K = 256, 128, 2048, 7168
g = torch.Generator(device="cuda").manual_seed(0)
def rf(shape):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)
a = rf((G, MMAX, K)); b = rf((G, N, K))
sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5; sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda"); ref = torch.zeros_like(out)
for mask_name, masked in {"full": torch.full((G,), MMAX, dtype=torch.int32, device="cuda"),
                          "random": torch.randint(0, MMAX + 1, (G,), dtype=torch.int32, device="cuda", generator=g)}.items():
    rows = int(masked.sum()); active = int((masked > 0).sum())
    alg = active * N * K + rows * (K + 4 * (K // 128) + 2 * N)
    variants = {}
    for name, (tile, waves, pp, st) in {"128x256 4w": ((128, 256), (2, 2), 0, 2), "128x256 4w 3st": ((128, 256), (2, 2), 0, 3),
                                    "128x256 8w": ((128, 256), (2, 4), 0, 2), "128x256 8w 3st": ((128, 256), (2, 4), 0, 3),
                                    "128x128 4w": ((128, 128), (2, 2), 0, 2), "128x128 4w 3st": ((128, 128), (2, 2), 0, 3),
                                    "64x256 4w 3st": ((64, 256), (1, 4), 0, 3)}.items():
        t = dga.tiling(MMAX, N, K, groups=G, expected_m=MMAX)
        t.m1, t.n1 = tile; t.wavesM, t.wavesN = waves; t.dispatchPolicyTag = pp; t.stages = st
        variants[name] = t
    ref.zero_(); dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), ref, masked, MMAX, tiling_=variants["128x256 4w"], sync=True)
    res = {k_: [] for k_ in variants}
    for name, t in variants.items():
        out.zero_(); dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, tiling_=t, sync=True)
        if not torch.equal(out.view(torch.int16), ref.view(torch.int16)): print("  !!", name, "differs from baseline")
    for rnd in range(3):
        for name, t in variants.items():
            dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, tiling_=t)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, tiling_=t)
            e1.record(); torch.cuda.synchronize()
            res[name].append(e0.elapsed_time(e1) * 200)
    for name, v in res.items():
        v = sorted(v); med = v[len(v) // 2]
        print(f"mask={mask_name} {name:18s}: median {med:.0f} us  {alg/med/1e3:.0f} GB/s  {2.0*N*K*rows/med/1e6:.0f} TF  {rows/med:.1f} Mtok/s", flush=True)
