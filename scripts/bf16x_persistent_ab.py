"""A/B of the bf16-exact policy's 128 x 256 build: persistent form (tiling.stages = 7) against one workgroup per tile (8), interleaved in
one process at sustained clocks -- BASELINE configs[1], configs[2], and configs[3] (masked grouped, full and random masks).
Usage: python scripts/bf16x_persistent_ab.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import deepgemm_ascend_amd as dga  # noqa: E402
from scripts.policy_perf import time_us  # noqa: E402


def tiling(m, n, k, persistent, groups=1):
    t = dga.tiling(m, n, k, groups=groups, expected_m=m, policy="bf16_exact") if groups > 1 else dga.tiling(m, n, k, policy="bf16_exact")
    t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.dispatchPolicyTag = 128, 256, 1, 0, 7
    t.stages, t.wavesM, t.wavesN = (7 if persistent else 8), 2, 4
    return t


def main():
    res = {}
    for name in ("dense_4096", "dsv3_prefill"):
        m, n, k = bench.WORKLOADS[name]
        a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        for rep in range(2):
            for pers in (False, True):
                t = tiling(m, n, k, pers)
                fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
                us = time_us(fn, 200, 400)
                res.setdefault(f"{name}_{'persistent' if pers else 'one_tile'}", []).append(round(us, 2))
                print(name, "persistent" if pers else "one tile per workgroup", f"{us:.2f} us  {2.0 * m * n * k / us / 1e6:.0f} TFLOP/s", flush=True)
    G, MMAX, N, K = 256, 128, 2048, 7168
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(0, 120, (G, MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
    sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5
    sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
    out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
    for mask, masked in (("full", torch.full((G,), MMAX, dtype=torch.int32, device="cuda")),
                         ("random", torch.randint(0, MMAX + 1, (G,), dtype=torch.int32, device="cuda", generator=g))):
        rows = int(masked.sum()); byt = G * N * K + rows * (K + 4 * (K // 128) + 2 * N)
        for rep in range(2):
            for pers in (False, True):
                t = tiling(MMAX, N, K, pers, groups=G)
                fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, tiling_=t, policy="bf16_exact")
                us = time_us(fn, 20, 100)
                res.setdefault(f"grouped_{mask}_{'persistent' if pers else 'one_tile'}", []).append(round(us, 1))
                print("grouped", mask, "persistent" if pers else "one tile per workgroup", f"{us:.1f} us  {byt / us / 8e6:.3f} of 8 TB/s", flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
