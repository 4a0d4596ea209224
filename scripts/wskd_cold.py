"""Cold timings of kernelSerial 6's two builds (LDS-DMA rings / registers) against the operator's own pick on short-M shapes.
Usage: python scripts/wskd_cold.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from scripts.wsk_cold import operand_sets, time_cold  # noqa: E402

SHAPES = [(1, 7168, 18432), (8, 7168, 18432), (16, 7168, 18432), (32, 7168, 18432), (8, 18432, 7168), (16, 18432, 7168), (32, 18432, 7168),
          (8, 4096, 7168), (16, 4096, 7168), (32, 4096, 7168), (16, 2048, 7168), (16, 7168, 2048), (16, 24576, 1536), (16, 32768, 512),
          (16, 7168, 16384), (16, 129280, 7168)]


def main():
    rows = []
    for m, n, k in SHAPES:
        sets = operand_sets(m, n, k)
        base = dga.tiling(m, n, k)
        cands = {"operator": base}
        for name, st in (("wsk_dma", 3), ("wsk_reg", 1)):
            t = dga.tiling(m, n, k); t.kernelSerial, t.splitkFactor, t.stages = 6, 1, st
            cands[name] = t
        res = {}
        for rep in range(2):
            for name, t in cands.items():
                fn = lambda s, t=t: dga.gemm_fp8_fp8_bf16_nt((s[0], s[1]), (s[2], s[3]), s[4], tiling_=t)
                res.setdefault(name, []).append(time_cold(fn, sets))
        byt = m * k + n * k + 2 * m * n
        row = {"shape": [m, n, k], "operator_pick": f"{base.m1}x{base.n1} serial {base.kernelSerial} split {base.splitkFactor}"}
        for name in cands:
            row[name + "_us"] = round(min(res[name]), 2)
        row["dma_vs_operator"] = round(row["wsk_dma_us"] / row["operator_us"], 3)
        row["dma_frac_of_8TBs"] = round(byt / row["wsk_dma_us"] / 8e6, 3)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
