"""per_token_cast_to_fp8 [32768, 7168] (and a decode-sized [128, 7168]): blocks per 16-lane group ($DGA_CAST_UNROLL = 1 / 2 / 4), one
subprocess each, device time by graph replay; the outputs of every variant are compared byte for byte with the one-block kernel's."""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys, hashlib; sys.path.insert(0, %r)
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
for rows, k in ((32768, 7168), (4096, 7168), (128, 7168)):
    for dt in (torch.bfloat16, torch.float32):
        g = torch.Generator(device="cuda").manual_seed(1)
        x = torch.randn((rows, k), device="cuda", generator=g).to(dt)
        q, sf = dga.per_token_cast_to_fp8(x)
        torch.cuda.synchronize()
        h = hashlib.sha1(q.view(torch.uint8).cpu().numpy().tobytes() + sf.cpu().numpy().tobytes()).hexdigest()[:12]
        us = sweep.graph_us(lambda: dga.per_token_cast_to_fp8(x), 10, 3, 30.0)
        byt = rows * k * (x.element_size() + 1) + rows * (k // 128) * 4
        print(rows, k, str(dt).split(".")[-1], "%%.1f" %% us, "%%.2f" %% (byt / us / 1e6), h, flush=True)
''' % str(ROOT)
for u in ("1", "2", "4"):
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, DGA_CAST_UNROLL=u), capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        print("unroll", u, line, flush=True)
    if r.returncode:
        print(r.stderr[-800:])
