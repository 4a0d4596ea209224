"""Does the 4096^3 rate depend on how long the GPU has been busy before the timed region? (development aid)"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
m = n = k = 4096
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
t = dga.tiling(m, n, k)
def run(steps):
    for _ in range(steps): dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
def timed(steps=200):
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); run(steps); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / steps
for idle_ms, warm in ((500, 20), (500, 200), (500, 2000), (500, 20000), (0, 20), (2000, 20), (2000, 2000)):
    torch.cuda.synchronize(); time.sleep(idle_ms / 1e3)
    run(warm)
    us = [timed() for _ in range(3)]
    print(f"idle {idle_ms} ms, warmup {warm} steps: 200 timed steps x3 = {us[0]:.1f} / {us[1]:.1f} / {us[2]:.1f} us", flush=True)
