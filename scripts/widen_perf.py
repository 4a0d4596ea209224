"""Device timing of the rows either side of the GEMM: the quantisers and the contiguous-grouped layout
(development aid; bench.py is the judged entry)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga


def timeit(fn, iters=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def rand_fp8(shape, g):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)


def casts():
    for dtype in (torch.bfloat16, torch.float32):
        for rows, k in ((4096, 7168), (32768, 7168), (128, 7168)):
            x = torch.randn((rows, k), device="cuda").to(dtype)
            us = timeit(lambda: dga.per_token_cast_to_fp8(x))
            byt = rows * k * (x.element_size() + 1) + rows * (k // 128) * 4
            print(f"per_token_cast {dtype} {rows}x{k}: {us:.1f} us  {byt / us / 1e3:.0f} GB/s", flush=True)
        for rows, k in ((2048, 7168), (7168, 2048)):
            x = torch.randn((rows, k), device="cuda").to(dtype)
            us = timeit(lambda: dga.per_block_cast_to_fp8(x))
            byt = rows * k * (x.element_size() + 1)
            print(f"per_block_cast {dtype} {rows}x{k}: {us:.1f} us  {byt / us / 1e3:.0f} GB/s", flush=True)


def contiguous():
    g = torch.Generator(device="cuda").manual_seed(0)
    for groups, per, n, k in ((8, 1024, 4096, 7168), (8, 1024, 7168, 2048), (32, 128, 4096, 7168), (32, 256, 7168, 2048),
                              (4, 8192, 4096, 7168)):
        msum = groups * per
        a = rand_fp8((msum, k), g); b = rand_fp8((groups, n, k), g)
        sfa = torch.rand((msum, k // 128), device="cuda") + 0.5
        sfb = torch.rand((groups, n // 128, k // 128), device="cuda") + 0.5
        idx = torch.arange(groups, device="cuda", dtype=torch.int32).repeat_interleave(per)
        out = torch.empty((msum, n), dtype=torch.bfloat16, device="cuda")
        t = dga.tiling(msum, n, k, groups=groups, contiguous=True)
        us = timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t), iters=20, warm=5)
        tf = 2.0 * msum * n * k / us / 1e6
        byt = msum * k + groups * n * k + msum * n * 2
        print(f"contiguous G={groups} x {per} rows, N={n} K={k}: tile {t.m1}x{t.n1} st{t.stages} raster {t.swizzleOffset}: "
              f"{us:.1f} us  {tf:.0f} TFLOP/s  {byt / us / 1e3:.0f} GB/s", flush=True)
        for raster in (1, 2, 4, 16):
            t.swizzleOffset = raster
            us = timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t), iters=20, warm=5)
            print(f"    raster {raster}: {us:.1f} us  {2.0 * msum * n * k / us / 1e6:.0f} TFLOP/s", flush=True)


if __name__ == "__main__":
    casts()
    contiguous()
