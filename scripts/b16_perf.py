import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (1024, 4096, 7168), (128, 4096, 7168)]:
    x = torch.randn((1, m, k), device="cuda").to(torch.bfloat16); y = torch.randn((1, k, n), device="cuda").to(torch.bfloat16)
    z = torch.empty((1, m, n), dtype=torch.float32, device="cuda")
    for _ in range(3): dga.run_mmad_rtc(x, y, z)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): dga.run_mmad_rtc(x, y, z)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    ref = (x[0].float() @ y[0].float())
    err = ((z[0] - ref).abs().max() / ref.abs().max()).item()
    print(f"run_mmad_rtc bf16 {m}x{n}x{k}: {us:.1f} us (incl. transpose + sync)  {2*m*n*k/us/1e6:.0f} TFLOP/s  rel err {err:.2e}", flush=True)
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 2048, 7168), (128, 4096, 7168)]:
    a = torch.randn((m, k), device="cuda").to(torch.bfloat16); b = torch.randn((n, k), device="cuda").to(torch.bfloat16)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    for _ in range(5): dga.catlass_dynamic_matmul(a, b.t(), out)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): dga.catlass_dynamic_matmul(a, b.t(), out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    print(f"catlass_dynamic_matmul bf16 NT {m}x{n}x{k}: {us:.1f} us  {2*m*n*k/us/1e6:.0f} TFLOP/s", flush=True)
