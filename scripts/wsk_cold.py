"""Cold timings (operand sets rotated past the Infinity Cache) of the one-launch workgroup split-K kernel (kernelSerial 6) against
the operator's own pick (the tuned tile + two-launch split-K where the selector splits) on short-M shapes.
Usage: python scripts/wsk_cold.py [--quick]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402

SHAPES = [(8, 18432, 7168), (64, 7168, 18432), (64, 4096, 7168), (128, 4096, 7168), (1, 18432, 7168), (16, 18432, 7168),
          (32, 18432, 7168), (64, 18432, 7168), (16, 7168, 18432), (32, 7168, 18432), (8, 4096, 7168), (16, 4096, 7168),
          (32, 4096, 7168), (8, 7168, 2048), (32, 7168, 2048), (64, 2048, 7168), (16, 24576, 1536), (64, 24576, 1536),
          (16, 32768, 512), (8, 7168, 16384), (64, 7168, 16384), (16, 129280, 7168)]


def operand_sets(m, n, k, budget=768 << 20):
    per = m * k + n * k + 2 * m * n
    sets = max(2, min(64, budget // max(per, 1) + 1))
    out = []
    gen = torch.Generator(device="cuda").manual_seed(1)
    kb, nb = (k + 127) // 128, (n + 127) // 128
    for _ in range(sets):
        a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
        b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
        out.append((a, torch.rand((m, kb), device="cuda") + 0.5, b, torch.rand((nb, kb), device="cuda") + 0.5,
                    torch.empty((m, n), dtype=torch.bfloat16, device="cuda")))
    return out


def time_cold(fn_of_set, sets, iters=60):
    for s in sets[:4]:
        fn_of_set(s)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fn_of_set(sets[i % len(sets)])
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    shapes = [SHAPES[i] for i in (0, 4, 5, 6, 8, 10, 21)] if "--quick" in sys.argv else SHAPES
    rows = []
    for m, n, k in shapes:
        sets = operand_sets(m, n, k)
        t0 = dga.tiling(m, n, k)
        picked = t0.kernelSerial
        base = dga.tiling(m, n, k)
        if base.kernelSerial == 6:     # the selector already takes the new kernel: time the previous pick beside it
            import os
            os.environ["DGA_NO_WSK_PICK"] = "1"
        tw = dga.tiling(m, n, k); tw.kernelSerial, tw.splitkFactor = 6, 1
        res = {}
        for rep in range(2):
            for name, t in (("operator", base), ("wsk", tw)):
                if name == "wsk" and m > 64:
                    continue
                fn = lambda s, t=t: dga.gemm_fp8_fp8_bf16_nt((s[0], s[1]), (s[2], s[3]), s[4], tiling_=t)
                res.setdefault(name, []).append(time_cold(fn, sets))
        byt = m * k + n * k + 2 * m * n
        row = {"shape": [m, n, k], "operator_pick": f"{base.m1}x{base.n1} serial {base.kernelSerial} split {base.splitkFactor}",
               "operator_us": round(min(res["operator"]), 2), "wsk_us": round(min(res["wsk"]), 2) if "wsk" in res else None,
               "hbm_floor_us_at_8TBs": round(byt / 8e6, 2), "selector_takes_wsk": picked == 6}
        if row["wsk_us"]:
            row["wsk_vs_operator"] = round(row["wsk_us"] / row["operator_us"], 3)
            row["wsk_frac_of_8TBs"] = round(byt / row["wsk_us"] / 8e6, 3)
        row["operator_frac_of_8TBs"] = round(byt / row["operator_us"] / 8e6, 3)
        rows.append(row)
        print(json.dumps(row), flush=True)
        del sets
        torch.cuda.empty_cache()
    print(json.dumps({"rows": rows}))


if __name__ == "__main__":
    main()
