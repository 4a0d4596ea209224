"""BASELINE configs[3] (256 x (M <= 128, K = 7168, N = 2048)) by mask class: the masked grouped GEMM alone, one process, sustained
clocks.  Per class the in-contract default (bf16_exact, hint = the class's mean rows and hint = m_max) and the fast policy.
Usage: python scripts/r06_grouped_masks.py [--quick] [--stages S ...]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import deepgemm_ascend_amd as dga  # noqa: E402

G, MMAX, N, K = 256, 128, 2048, 7168


def masks():
    cpu = torch.Generator().manual_seed(99)
    r = lambda lo, hi: torch.randint(lo, hi + 1, (G,), generator=cpu).to(torch.int32).cuda()
    return [("full", torch.full((G,), MMAX, dtype=torch.int32, device="cuda")),
            ("random_0_128", r(0, 128)), ("random_65_128", r(65, 128)), ("random_0_64", r(0, 64)),
            ("random_0_32", r(0, 32)), ("random_0_16", r(0, 16)),
            ("const_96", torch.full((G,), 96, dtype=torch.int32, device="cuda")),
            ("const_80", torch.full((G,), 80, dtype=torch.int32, device="cuda")),
            ("const_64", torch.full((G,), 64, dtype=torch.int32, device="cuda")),
            ("const_48", torch.full((G,), 48, dtype=torch.int32, device="cuda")),
            ("const_16", torch.full((G,), 16, dtype=torch.int32, device="cuda"))]


def main():
    quick = "--quick" in sys.argv
    a, sfa, b, sfb = bench.make_grouped_inputs(G, MMAX, N, K, seed=0) if hasattr(bench, "make_grouped_inputs") else (None,) * 4
    if a is None:
        g = torch.Generator(device="cuda").manual_seed(0)
        a = torch.randint(0, 120, (G, MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
        b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
        sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5
        sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
    out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
    res = {}
    for name, masked in masks():
        rows = int(masked.sum())
        mean = max(1, rows // G)
        byt = int((masked > 0).sum()) * N * K + rows * (K + 4 * (K // 128) + 2 * N) + G * (N // 128) * (K // 128) * 4
        row = {"rows": rows, "bytes": byt}
        # the default call (the selector names the layout's own kernel, build 9), the persistent kernel it replaced (build 7) and the
        # tile the round-5 selector took from the hint (32 x 128 / 64 x 256 below 64 expected rows), the fast policy beside them
        legs = [("bf16_exact", dict(policy="bf16_exact"), mean, 0), ("bf16_exact_build_7_persistent", dict(policy="bf16_exact"), MMAX, 7),
                ("fast", dict(policy="fast"), mean, 0)]
        if mean <= 64:
            legs.insert(2, ("bf16_exact_r05_hint_tile", dict(policy="bf16_exact", tile=(32, 128) if mean <= 32 else (64, 256)), mean, 0))
        if "--base-only" in sys.argv:
            legs = legs[:2]
        for st in [int(x) for x in sys.argv[sys.argv.index("--builds") + 1:] if x.isdigit()] if "--builds" in sys.argv else []:
            legs.insert(0, (f"bf16_exact_build_{st}", dict(policy="bf16_exact"), MMAX, st))
        if "--knobs" in sys.argv:   # $DGA_BXG_KNOB values of the grouped kernel, A/B in this process
            for kn in sys.argv[sys.argv.index("--knobs") + 1].split(","):
                legs.insert(0, (f"bf16_exact_build_9_knob_{kn}", dict(policy="bf16_exact", knob=kn), MMAX, 9))
        import os
        for leg, kw, hint, st in legs:
            os.environ.pop("DGA_BXG_KNOB", None)
            if "knob" in kw:
                os.environ["DGA_BXG_KNOB"] = kw["knob"]
            t = dga.tiling(MMAX, N, K, groups=G, expected_m=hint, policy=kw["policy"])
            if st:
                t.build = st
            if "tile" in kw:
                t.m1, t.n1 = kw["tile"]; t.build = 0
                t.blockDim = G * -(-MMAX // t.m1) * -(-N // t.n1)
            fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, hint, tiling_=t)
            fn(); torch.cuda.synchronize()
            us = min(bench._prewarmed_us(fn, 20 if quick else 40, 60.0 if quick else 150.0) for _ in range(2))
            row[leg] = {"us": round(us, 1), "TBps": round(byt / us / 1e6, 3), "tile": f"{t.m1}x{t.n1}", "build": int(t.build),
                        "tok_per_s": round(rows / us * 1e6)}
            print(name, leg, row[leg], flush=True)
        res[name] = row
    print(json.dumps(res))


if __name__ == "__main__":
    main()
