"""Random-shape sweep of the 16-bit entry points against an fp32 matmul of the up-cast inputs (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch
import deepgemm_ascend_amd as dga
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 150):
    m = int(rng.choice([1, 5, 9, 16, 17, 40, 64, 100, 128, 257, 512, 1000, 2048, 4096])); n = int(rng.choice([1, 7, 8, 16, 72, 128, 200, 512, 1024, 2048, 4100, 8200, 20000]))
    k = int(rng.choice([3, 16, 64, 72, 128, 192, 512, 576, 1000, 1024, 4096, 8192, 16384])); dt = [torch.bfloat16, torch.float16][it & 1]
    if m * n * k > (1 << 34): continue
    a = (torch.randn((m, k), device="cuda") * 0.5).to(dt); b = (torch.randn((n, k), device="cuda") * 0.5).to(dt)
    gold = a.float() @ b.float().T
    out = torch.full((m, n), float("nan"), dtype=dt, device="cuda")
    dga.catlass_dynamic_matmul(a, b.t(), out, sync=True)
    tol = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * gold.abs() + 2e-3 * (k ** 0.5)
    r1 = float(((out.float() - gold).abs() > tol).float().mean())
    z = torch.full((1, m, n), float("nan"), dtype=torch.float32, device="cuda")
    dga.run_mmad_rtc(a[None], b.t().contiguous()[None], z)
    r2 = float(((z[0] - gold).abs() > 2e-4 * gold.abs() + 1e-4 * (k ** 0.5)).float().mean())
    if r1 > 1e-4 or r2 > 1e-4 or not torch.isfinite(out.float()).all() or not torch.isfinite(z).all():
        bad += 1; print("MISMATCH", m, n, k, dt, r1, r2, flush=True)
print("done, mismatching cases:", bad)
