"""Decode rows of the 16-bit operator (bf16), cold: the one-launch workgroup split-K (csrc/gemm_b16_wsk_kernel.hpp) against the
operator's planned tile kernel (+ combine where it splits).  Device time by graph replay over operand sets rotated past the Infinity
Cache.  Usage: python scripts/op16_wsk_cold.py"""
import json
import os
os.environ.setdefault("DGA_B16_DEV", "1")   # the 16-bit operators read their development switches per call only when told so (dga_b16.hip)

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

NK = [(576, 7168), (1536, 7168), (2048, 7168), (4096, 4096), (4096, 7168), (4096, 14336), (7168, 2048), (7168, 4608), (7168, 16384),
      (7168, 18432), (8192, 8192), (10240, 8192), (16384, 7168), (18432, 7168), (24576, 1536), (28672, 4096), (32768, 512), (57344, 8192),
      (129280, 7168)]


def main():
    for n, k in NK:
        for m in (1, 8, 16):
            per = 2 * (m * k + n * k + m * n)
            nset = max(2, min(16, (320 << 20) // per + 1))
            g = torch.Generator(device="cuda").manual_seed(n + k + m)
            sets = [((torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                     (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                     torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(nset)]
            row = {"shape": [m, n, k]}
            for name, env in (("tile_us", "0"), ("wsk_us", "1")):
                os.environ["DGA_B16_WSK"] = env
                turn = [0]
                def fn():
                    x, w, o = sets[turn[0] % nset]; turn[0] += 1
                    dga.catlass_dynamic_matmul(x, w.t(), o)
                n_it = nset * max(1, 16 // nset)
                row[name] = round(min(u for u in (sweep.graph_us(fn, n_it, replays=3) for _ in range(2)) if u), 2)
            os.environ.pop("DGA_B16_WSK", None)
            row["ratio"] = round(row["wsk_us"] / row["tile_us"], 3)
            row["wsk_frac_of_8TBs"] = round(per / row["wsk_us"] / 8e6, 3)
            print(json.dumps(row), flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
