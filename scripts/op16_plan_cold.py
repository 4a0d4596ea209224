"""The 16-bit operator (bf16) at 1..128 rows, cold: the plan the operator picks (auto_us; rule_us = without its table of swept
plans) against every (tile, split-K) the menu offers and, at 16 rows or fewer, the one-launch workgroup split-K ($DGA_B16_PLAN /
$DGA_B16_WSK / $DGA_B16_NO_TABLE are read per call).  Device time by graph replay over operand sets rotated past the Infinity Cache.
--mmad: the same sweep through run_mmad_rtc / run_mmad_bench's launch (fp16 x[M,K], y[K,N] read where it lies, f32 out).
Usage: python scripts/op16_plan_cold.py [--mmad] [m ...]"""
import json
import os
os.environ.setdefault("DGA_B16_DEV", "1")   # the 16-bit operators read their development switches per call only when told so (dga_b16.hip)

import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd import _lib, api  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

NK = [(4096, 16384), (2048, 32768), (576, 7168), (1536, 7168), (2048, 7168), (4096, 4096), (4096, 7168), (4096, 14336), (7168, 2048), (7168, 4608), (7168, 16384),
      (7168, 18432), (8192, 8192), (10240, 8192), (16384, 7168), (18432, 7168), (24576, 1536), (28672, 4096), (32768, 512), (57344, 8192),
      (129280, 7168), (2112, 7168), (4608, 7168), (6144, 4096), (7168, 1536), (8192, 28672), (14336, 4096), (5120, 5120), (13824, 5120),
      (5120, 13824), (3072, 8192), (1024, 4096), (16384, 16384), (53248, 16384), (16384, 53248), (27648, 5120), (5120, 27648), (3584, 3584),
      (37888, 3584), (3584, 18944)]
TILES = [(16, 128), (32, 128), (64, 128), (128, 128), (128, 256)]
SPLITS = [1, 2, 3, 4, 6, 8, 12, 16]


def main():
    mmad = "--mmad" in sys.argv
    ms = [int(a) for a in sys.argv[1:] if a != "--mmad"] or [8, 16, 32, 64, 128]
    lib = _lib.lib()
    for n, k in NK:
        for m in ms:
            per = 2 * (m * k + n * k + m * n)
            nset = max(2, min(16, (320 << 20) // per + 1))
            g = torch.Generator(device="cuda").manual_seed(n + k + m)
            if mmad:
                sets = [((torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.float16),
                         (torch.randn((k, n), device="cuda", generator=g) * 0.5).to(torch.float16),
                         torch.empty((m, n), dtype=torch.float32, device="cuda")) for _ in range(nset)]
            else:
                sets = [((torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                         (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(torch.bfloat16),
                         torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(nset)]
            n_it = nset * max(1, 16 // nset)

            def timed():
                turn = [0]
                def fn():
                    x, w, o = sets[turn[0] % nset]; turn[0] += 1
                    if mmad:
                        ws_ptr, ws_bytes = api._mmad_workspace(1, m, n, k, x)
                        rc = lib.dga_run_mmad_rtc_ws(x.data_ptr(), w.data_ptr(), o.data_ptr(), 1, m, n, k, api._dt16(x), ws_ptr, ws_bytes,
                                                     api._stream_ptr(o))
                        assert rc == 0, rc
                        return
                    dga.catlass_dynamic_matmul(x, w.t(), o)
                return min(u for u in (sweep.graph_us(fn, n_it, replays=3) for _ in range(2)) if u)

            for e in ("DGA_B16_PLAN", "DGA_B16_WSK", "DGA_B16_NO_TABLE"):
                os.environ.pop(e, None)
            row = {"shape": [m, n, k], "auto_us": round(timed(), 2), "plans": {}}
            os.environ["DGA_B16_NO_TABLE"] = "1"
            row["rule_us"] = round(timed(), 2)
            if m <= 32 and not mmad:
                os.environ["DGA_B16_WSK"] = "1"
                row["plans"]["wsk"] = round(timed(), 2)
            os.environ["DGA_B16_WSK"] = "0"
            for bm, bn in TILES:
                if (bm == 32 and m > 64) or (bm == 16 and m > 32):
                    continue
                for s in SPLITS:
                    if s > 1 and (k // 64) // s < 4:
                        continue
                    if s * m * n * 4 > (1 << 30):
                        continue
                    os.environ["DGA_B16_PLAN"] = f"{bm},{bn},{s}"
                    row["plans"][f"{bm}x{bn}/{s}"] = round(timed(), 2)
            for e in ("DGA_B16_PLAN", "DGA_B16_WSK", "DGA_B16_NO_TABLE"):
                os.environ.pop(e, None)
            best = min(row["plans"], key=row["plans"].get)
            row["best"] = best
            row["best_us"] = row["plans"][best]
            row["auto_over_best"] = round(row["auto_us"] / row["best_us"], 3)
            row["rule_over_best"] = round(row["rule_us"] / row["best_us"], 3)
            print(json.dumps(row), flush=True)
            del sets
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
