"""The four-wave 32x32x16 build of the 16-bit 256x256 tile (csrc/gemm_b16_w4_kernel.hpp; selectable: $DGA_B16_W4=1 routes the
operator's unsplit 256x256 plans to it): same k order per output as the 8-wave build, so the same bytes -- on full rasters, ragged
edges and short K, bf16 and fp16.  Reference counterpart: the operator's device entry
(/root/reference/aclnn_catlass_dynamic_matmul/op_kernel/catlass_dynamic_matmul.cpp:16-45)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

CHILD = r'''
import os, sys
sys.path.insert(0, %r)
import torch
import deepgemm_ascend_amd as dga
os.environ["DGA_B16_PLAN"] = "256,256,1"
for dt in (torch.bfloat16, torch.float16):
    for (m, n, k) in [(512, 768, 256), (700, 1000, 192), (2100, 4104, 1024), (256, 256, 64), (4096, 4096, 512)]:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        x = (torch.randn((m, k), device="cuda", generator=g) * 0.5).to(dt)
        w = (torch.randn((n, k), device="cuda", generator=g) * 0.5).to(dt)
        o = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
        dga.catlass_dynamic_matmul(x, w.t(), o, sync=True)
        print(m, n, k, str(dt), o.view(torch.int16).to(torch.int64).sum().item(), int(torch.isnan(o.float()).sum()),
              o.view(torch.int16)[::7, ::5].to(torch.int64).mul(torch.arange(o[::7, ::5].numel(), device="cuda").view(o[::7, ::5].shape) %% 1009).sum().item())
''' % str(ROOT)


def _run(w4):
    env = dict(os.environ)
    env.pop("DGA_B16_W4", None)
    if w4:
        env["DGA_B16_W4"] = "1"     # (read once per process)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return [l for l in r.stdout.splitlines() if l and l[0].isdigit()]


def test_same_bytes_as_the_eight_wave_build():
    a, b = _run(True), _run(False)
    assert len(a) == 10 and a == b, (a, b)
    assert all(l.split()[5] == "0" for l in a)      # no NaN left of the output's initial fill
