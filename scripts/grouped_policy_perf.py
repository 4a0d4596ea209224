"""BASELINE configs[3] (256 x (M <= 128, K = 7168, N = 2048)) under each arithmetic policy: the masked grouped GEMM alone, full and
random masks, in one process at sustained clocks.  Usage: python scripts/grouped_policy_perf.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from scripts.policy_perf import time_us  # noqa: E402

G, MMAX, N, K = 256, 128, 2048, 7168


def main():
    g = torch.Generator(device="cuda").manual_seed(0)
    a = torch.randint(0, 120, (G, MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
    sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5
    sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
    out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
    res = {}
    for mask in ("full", "random"):
        masked = (torch.full((G,), MMAX, dtype=torch.int32, device="cuda") if mask == "full" else
                  torch.randint(0, MMAX + 1, (G,), dtype=torch.int32, device="cuda", generator=g))
        rows = int(masked.sum())
        byt = G * N * K + rows * (K + 4 * (K // 128) + 2 * N) + G * (N // 128) * (K // 128) * 4
        for pol in ("fast", "bf16_exact"):
            kw = {} if pol == "fast" else {"policy": pol}
            fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, **kw)
            us = time_us(fn, 40, 300)
            res[f"{mask}_{pol}"] = {"us": round(us, 1), "TBps": round(byt / us / 1e6, 3), "frac_of_8TBps": round(byt / us / 8e6, 4),
                                    "tok_per_s": round(rows / us * 1e6, 0)}
            print(mask, pol, res[f"{mask}_{pol}"], flush=True)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
