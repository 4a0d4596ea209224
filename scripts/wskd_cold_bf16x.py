"""The bf16-exact policy on decode rows, cold: that policy's own tiling (tile kernel, two-launch split-K where it splits) against the
workgroup split-K on LDS-DMA rings with the same arithmetic (kernelSerial 6 + dispatchPolicyTag 7).  Device time by graph replay over
operand sets rotated past the Infinity Cache.  Usage: python scripts/wskd_cold_bf16x.py [shapes file]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402
from scripts.wsk_cold import operand_sets  # noqa: E402


def main():
    f = sys.argv[1] if len(sys.argv) > 1 else str(ROOT / "scripts" / "decode_grid_shapes_m32.txt")
    shapes = [tuple(int(x) for x in s.split(",")) for s in Path(f).read_text().split()]
    for m, n, k in shapes:
        sets = operand_sets(m, n, k, budget=512 << 20)
        base = dga.tiling(m, n, k, policy="bf16_exact")
        t6 = dga.tiling(m, n, k, policy="bf16_exact")
        t6.kernelSerial, t6.splitkFactor, t6.stages, t6.m1, t6.n1 = 6, 1, 3, (16 if m <= 16 else 32), 128
        row = {"shape": [m, n, k], "pick": f"{base.m1}x{base.n1} serial {base.kernelSerial} split {base.splitkFactor}"}
        for name, t in (("pick_us", base), ("wsk_us", t6)):
            turn = [0]
            def fn(t=t):
                s = sets[turn[0] % len(sets)]
                turn[0] += 1
                dga.gemm_fp8_fp8_bf16_nt((s[0], s[1]), (s[2], s[3]), s[4], tiling_=t, policy="bf16_exact")
            n_it = len(sets) * max(1, 24 // len(sets))
            best = None
            for _ in range(2):
                turn[0] = 0
                us = sweep.graph_us(fn, n_it, replays=3)
                best = us if best is None or (us is not None and us < best) else best
            row[name] = round(best, 2) if best else None
        if row["pick_us"] and row["wsk_us"]:
            row["ratio"] = round(row["wsk_us"] / row["pick_us"], 3)
        print(json.dumps(row), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
