"""The activation quantiser (per_token_cast_to_fp8: bf16 -> e4m3fn + 1x128 f32 scales) at decode-time row counts: device time by graph
replay.  Usage: python scripts/cast_small_rows.py"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402

for k in (2048, 7168, 18432):
    for rows in (1, 8, 64, 512, 4096, 32768):
        x = torch.randn((rows, k), device="cuda", dtype=torch.bfloat16)
        fn = lambda: dga.per_token_cast_to_fp8(x)
        fn(); torch.cuda.synchronize()
        us = min(sweep.graph_us(fn, 20, replays=3, prewarm_ms=20.0) for _ in range(2))
        byt = rows * k * 3 + rows * (k // 128) * 4
        print(json.dumps({"rows": rows, "k": k, "us": round(us, 2), "gbps": round(byt / us / 1e3, 1)}), flush=True)
