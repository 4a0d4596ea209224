"""The in-contract (bf16-exact) policy on masked grouped problems it was never tuned on: every tile of the policy's menu (one-tile and
persistent form of the 128x256 build) timed at a full mask and a ragged one against the tiling dga_tiling(policy="bf16_exact")
names.  Usage: python scripts/grouped_bf16x_regret.py"""
import json
import math
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

SHAPES = [(64, 16, 5120, 5120), (64, 128, 5120, 5120), (16, 64, 3072, 8192), (128, 32, 3072, 8192), (64, 64, 1536, 4096), (32, 128, 4096, 7168),
          (256, 32, 2048, 7168), (128, 128, 7168, 2048)]
MENU = [(128, 256, 3), (128, 256, 7), (128, 256, 8), (128, 128, 3), (64, 256, 3), (64, 128, 3), (32, 128, 3)]   # (m1, n1, stages: 7 / 8 = persistent / one-tile forced)


def main():
    reg = []
    for (G, mm, n, k) in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(G + mm + n)
        a = torch.randint(0, 120, (G, mm, k), dtype=torch.uint8, device="cuda", generator=g)
        b = torch.randint(0, 120, (G, n, k), dtype=torch.uint8, device="cuda", generator=g)
        sfa = torch.rand((G, mm, -(-k // 128)), device="cuda", generator=g) + 0.5
        sfb = torch.rand((G, -(-n // 128), -(-k // 128)), device="cuda", generator=g) + 0.5
        out = torch.zeros((G, mm, n), dtype=torch.bfloat16, device="cuda")
        for mask_name, masked in (("full", torch.full((G,), mm, dtype=torch.int32, device="cuda")),
                                  ("ragged", torch.randint(0, mm + 1, (G,), dtype=torch.int32, device="cuda", generator=g))):
            em = int(masked.max())
            res = {}
            t0 = dga.tiling(mm, n, k, groups=G, expected_m=em, policy="bf16_exact")
            fn = lambda t=t0: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, em, tiling_=t, policy="bf16_exact")
            fn(); torch.cuda.synchronize()
            ref = out.clone()
            res["auto"] = min(sweep.graph_us(fn, 4, replays=3, prewarm_ms=30.0) for _ in range(2))
            for (m1, n1, st) in MENU:
                if m1 < min(mm, 128) and m1 < em:      # a tile shorter than the rows of a group streams the weights again per tile row
                    continue
                t = dga.tiling(mm, n, k, groups=G, expected_m=em, policy="bf16_exact")
                t.m1, t.n1, t.stages, t.wavesM, t.wavesN = m1, n1, st, 0, 0
                f2 = lambda t=t: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, em, tiling_=t, policy="bf16_exact")
                try:
                    out.zero_(); f2(); torch.cuda.synchronize()
                except Exception:   # noqa: BLE001
                    continue
                if not torch.equal(out.view(torch.int16), ref.view(torch.int16)):
                    continue                             # (tiles of another height order the sums differently only if rows differ: skip mismatches)
                res[f"{m1}x{n1}/st{st}"] = min(sweep.graph_us(f2, 4, replays=3, prewarm_ms=30.0) for _ in range(2))
            if len(res) < 2:
                continue
            best = min((p for p in res if p != "auto"), key=res.get)
            reg.append(res["auto"] / res[best])
            print(json.dumps({"shape": [G, mm, n, k], "mask": mask_name, "pick": f"{t0.m1}x{t0.n1}/st{t0.stages}", "auto_us": round(res["auto"], 1), "best": best,
                              "best_us": round(res[best], 1), "ratio": round(reg[-1], 3)}), flush=True)
    print("cases", len(reg), "geomean", round(math.exp(sum(math.log(r) for r in reg) / len(reg)), 4), "max", round(max(reg), 3))


if __name__ == "__main__":
    main()
