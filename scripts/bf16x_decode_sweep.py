"""The bf16-exact policy's own tiling on the short-M rows (64 / 128 rows) against every (tile, split-K) of its menu, cold: how far is
dga_tiling_bf16_exact's pick from the best candidate?  Usage: python scripts/bf16x_decode_sweep.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402
from scripts.wsk_cold import operand_sets  # noqa: E402

SHAPES = [(64, 4096, 7168), (64, 7168, 18432), (64, 18432, 7168), (64, 24576, 1536), (64, 32768, 512), (64, 7168, 16384),
          (128, 4096, 7168), (128, 7168, 18432), (128, 18432, 7168), (48, 4096, 7168), (96, 7168, 2048), (32, 18432, 7168)]
TILES = [(128, 256), (128, 128), (64, 256), (64, 128), (32, 128)]


def main():
    for m, n, k in SHAPES:
        sets = operand_sets(m, n, k, budget=512 << 20)
        kb = (k + 127) // 128
        pick = dga.tiling(m, n, k, policy="bf16_exact")
        res = {}
        def timed(t):
            turn = [0]
            def fn():
                s = sets[turn[0] % len(sets)]; turn[0] += 1
                dga.gemm_fp8_fp8_bf16_nt((s[0], s[1]), (s[2], s[3]), s[4], tiling_=t, policy="bf16_exact")
            n_it = len(sets) * max(1, 16 // len(sets))
            return min(x for x in (sweep.graph_us(fn, n_it, replays=3) for _ in range(2)) if x)
        res["pick"] = timed(pick)
        for bm, bn in TILES:
            if bm > max(32, 2 * m):
                continue
            for sk in (1, 2, 3, 4, 6, 8, 12):
                if sk > 1 and kb < 4 * sk:
                    continue
                t = dga.tiling(m, n, k, policy="bf16_exact")
                t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.stages, t.wavesM, t.wavesN = bm, bn, sk, (4 if sk > 1 else 0), 3, 0, 0
                try:
                    res[f"{bm}x{bn}s{sk}"] = timed(t)
                except Exception as e:
                    res[f"{bm}x{bn}s{sk}"] = None
        ok = {k2: v for k2, v in res.items() if v}
        best = min(ok, key=ok.get)
        print(json.dumps({"shape": [m, n, k], "pick": f"{pick.m1}x{pick.n1}s{pick.splitkFactor} serial {pick.kernelSerial}", "pick_us": round(res["pick"], 2),
                          "best": best, "best_us": round(ok[best], 2), "regret": round(res["pick"] / ok[best], 3),
                          "top": {k2: round(v, 2) for k2, v in sorted(ok.items(), key=lambda kv: kv[1])[:4]}}), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
