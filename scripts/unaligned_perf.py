"""The odd-K shapes of the reference's list: the fused form (kernelSerial 2: loader waves read the rows in place) against the
operator's pick (padding pass + tuned tile) and against the padding pass in front of the same 128x256 loader-wave tile.
Usage: python scripts/unaligned_perf.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from scripts.policy_perf import time_us  # noqa: E402

SHAPES = [(1279, 5003, 7681), (3511, 6151, 8191), (5119, 6997, 9901), (2048, 4096, 7169), (512, 4096, 7000)]


def main():
    rows = []
    for m, n, k in SHAPES:
        gen = torch.Generator(device="cuda").manual_seed(k)
        a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
        b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
        kb = (k + 127) // 128
        sfa = torch.rand((m, kb), device="cuda") + 0.5; sfb = torch.rand(((n + 127) // 128, kb), device="cuda") + 0.5
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        t_op = dga.tiling(m, n, k)
        if t_op.kernelSerial == 2:   # the selector already takes the fused form: its previous pick is the padding pass + this tile
            t_op.kernelSerial = 0
        t_pad = dga.tiling(m, n, k); t_pad.m1, t_pad.n1, t_pad.stages, t_pad.wavesM, t_pad.wavesN, t_pad.splitkFactor = 128, 256, 3, 2, 2, 1
        t_pad.dispatchPolicyTag, t_pad.kernelSerial = 4, 0
        t_fu = dga.tiling(m, n, k); t_fu.m1, t_fu.n1, t_fu.stages, t_fu.wavesM, t_fu.wavesN, t_fu.splitkFactor = 128, 256, 3, 2, 2, 1
        t_fu.dispatchPolicyTag, t_fu.kernelSerial = 4, 2
        res = {}
        for rep in range(2):
            for name, t in (("operator", t_op), ("pad_then_128x256_loaders", t_pad), ("fused", t_fu)):
                fn = lambda t=t: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
                res.setdefault(name, []).append(time_us(fn, 100, 300))
        row = {"shape": [m, n, k], "operator_pick": f"{t_op.m1}x{t_op.n1} policy {t_op.dispatchPolicyTag} split {t_op.splitkFactor}",
               **{k2: round(min(v), 1) for k2, v in res.items()}}
        row["fused_vs_operator"] = round(row["fused"] / row["operator"], 3)
        row["tflops_fused"] = round(2.0 * m * n * k / row["fused"] / 1e6, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({"rows": rows}))


if __name__ == "__main__":
    main()
