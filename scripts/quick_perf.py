"""Quick device timing of the fp8 GEMM variants (development aid; bench.py is the judged entry)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga


def rand_fp8(shape, g):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)


def time_gemm(m, n, k, variant=None, iters=50, warm=10):
    g = torch.Generator(device="cuda").manual_seed(0)
    a = rand_fp8((m, k), g); b = rand_fp8((n, k), g)
    sfa = torch.rand((m, (k + 127) // 128), device="cuda") + 0.5
    sfb = torch.rand(((n + 127) // 128, (k + 127) // 128), device="cuda") + 0.5
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    if variant:
        t.m1, t.n1 = variant
    for _ in range(warm):
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    tf = 2.0 * m * n * k / us / 1e6
    return us, tf, (t.m1, t.n1)


if __name__ == "__main__":
    shapes = [(4096, 4096, 4096), (4096, 2048, 7168), (8192, 8192, 8192), (1024, 4096, 7168), (128, 4096, 7168)]
    for (m, n, k) in shapes:
        for var in [None, (256, 256), (128, 256), (256, 128), (128, 128)]:
            try:
                us, tf, v = time_gemm(m, n, k, var)
                print(f"{m}x{n}x{k} variant={v} {'(auto)' if var is None else ''}: {us:.1f} us  {tf:.0f} TFLOP/s", flush=True)
            except Exception as e:
                print(m, n, k, var, "ERR", e, flush=True)
