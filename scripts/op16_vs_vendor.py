"""The 16-bit operator (catlass_dynamic_matmul, bf16, NT) beside the vendor GEMM library as torch.matmul reaches it (hipBLASLt /
Tensile), on the reference's 18-shape list: device time by graph replay, warm (--cold: at 256 rows or fewer the operands rotate through > 256 MB of copies, as in a model whose layers
do not fit the Infinity Cache).  A yardstick only -- nothing in the product calls a GEMM library.
Usage: python scripts/op16_vs_vendor.py [--cold]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402


def main():
    cold = "--cold" in sys.argv
    for (m, n, k) in sweep.SHAPE_GROUP:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        nset = max(2, min(16, (320 << 20) // (2 * (m * k + n * k + m * n)) + 1)) if cold and m <= 256 else 1
        sets = [((torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                 (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                 torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(nset)]
        turn = [0]

        def ours():
            x, w, o = sets[turn[0] % nset]; turn[0] += 1
            dga.catlass_dynamic_matmul(x, w.t(), o)

        def vendor():
            x, w, o = sets[turn[0] % nset]; turn[0] += 1
            torch.matmul(x, w.t(), out=o)
        ours(); vendor(); torch.cuda.synchronize()
        t = {}
        n_it = nset * max(1, 10 // nset)
        for rnd in range(3):   # interleaved: both see the same clocks
            for name, fn in (("ours_us", ours), ("vendor_us", vendor)):
                turn[0] = 0
                u = sweep.graph_us(fn, n_it, replays=3, prewarm_ms=30.0)
                if u:
                    t[name] = min(t.get(name, 1e30), u)
        row = {"shape": [m, n, k], "cold": nset > 1, **{a: round(b, 2) for a, b in t.items()}}
        if len(t) == 2:
            row["ours_over_vendor"] = round(t["ours_us"] / t["vendor_us"], 3)
        print(json.dumps(row), flush=True)


if __name__ == "__main__":
    main()
