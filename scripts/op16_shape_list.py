"""The reference's 18-shape sweep list (framework/benchmark/benchmark.py:24-44) through the reference's OWN operator slot -- the
16-bit catlass_dynamic_matmul (bf16 in, bf16 out, NT) -- warm, device time by graph replay, with the bound that applies (dense bf16
matrix peak 2.5 PFLOP/s, HBM 8 TB/s) and the fraction reached.  Usage: python scripts/op16_shape_list.py [--cold]"""
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
        nset = max(1, min(16, (320 << 20) // (2 * (m * k + n * k) + 1) + 1)) if cold and m <= 256 else 1
        sets = [((torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                 (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                 torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(nset)]
        turn = [0]
        def fn():
            x, w, o = sets[turn[0] % nset]; turn[0] += 1
            dga.catlass_dynamic_matmul(x, w.t(), o)
        fn(); torch.cuda.synchronize()
        x, w, o = sets[0]
        ref = x.float() @ w.float().t()
        err = float((o.float() - ref).abs().max() / ref.abs().max())
        n_it = nset * max(1, 16 // nset)
        us = min(u for u in (sweep.graph_us(fn, n_it, replays=3) for _ in range(2)) if u)
        flops, byt = 2.0 * m * n * k, 2.0 * (m * k + n * k + m * n)
        t_m, t_h = flops / 2.5e9, byt / 8e6
        print(json.dumps({"m": m, "n": n, "k": k, "us": round(us, 2), "tflops": round(flops / us / 1e6, 1), "gbps": round(byt / us / 1e3, 1),
                          "bound": "mfma" if t_m >= t_h else "hbm", "frac": round(max(t_m, t_h) / us, 3), "rel_err": round(err, 5),
                          "cold": bool(cold and m <= 256)}), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
