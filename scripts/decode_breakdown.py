"""Where a cold short-M call spends its time: run under `rocprofv3 --kernel-trace --stats` -- per-kernel durations of the tile
kernel and the split-K combine for M = 16 / 64 / 128 on two weight shapes, operand sets rotated past the Infinity Cache.
Usage: rocprofv3 --kernel-trace --stats -d gpurun_out/prof_decode -- python3 scripts/decode_breakdown.py [M ...]"""
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from scripts.wsk_cold import operand_sets, time_cold  # noqa: E402


def main():
    ms = [int(x) for x in sys.argv[1:]] or [16, 64, 128]
    for n, k in ((7168, 18432), (4096, 7168)):
        for m in ms:
            sets = operand_sets(m, n, k)
            picks = [dga.tiling(m, n, k)]
            if picks[0].kernelSerial == 6:   # beside the one-launch pick: the two-launch split-K it replaced (16x128 tiles, loader waves)
                t2 = dga.tiling(m, n, k)
                t2.kernelSerial, t2.splitkFactor, t2.m1, t2.n1, t2.stages, t2.dispatchPolicyTag = 4, (4 if k > 8192 else 8), 16, 128, 3, 4
                picks.append(t2)
            for t in picks:
                fn = lambda s: dga.gemm_fp8_fp8_bf16_nt((s[0], s[1]), (s[2], s[3]), s[4], tiling_=t)
                us = time_cold(fn, sets, iters=40)
                print(f"M={m} N={n} K={k}: {t.m1}x{t.n1} split {t.splitkFactor} serial {t.kernelSerial} policy {t.dispatchPolicyTag} "
                      f"stages {t.stages}: {us:.2f} us per call (events, {len(sets)} sets)", flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
