"""Indexed masked GEMM (token rows gathered where they lie) against the packed masked GEMM on BASELINE configs[3], one
process, policies 4 and 5 (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from widen_perf import timeit

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(5)
G, m_max, n, k = 256, 128, 2048, 7168
kb = k // 128
T = G * m_max
tok = torch.randint(0, 120, (T, k), dtype=torch.uint8, device=dev, generator=gen)
tsf = torch.rand((T, kb), device=dev, generator=gen) + 0.5
b = torch.randint(0, 120, (G, n, k), dtype=torch.uint8, device=dev, generator=gen)
sfb = torch.rand((G, n // 128, kb), device=dev, generator=gen) + 0.5
mm = torch.full((G,), m_max, dtype=torch.int32, device=dev)
out_rows = torch.empty((T, n), dtype=torch.bfloat16, device=dev)
out_packed = torch.empty((G, m_max, n), dtype=torch.bfloat16, device=dev)
for order in ("identity", "random"):
    idx = torch.arange(T, device=dev) if order == "identity" else torch.randperm(T, device=dev, generator=gen)
    idx = idx.to(torch.int64).contiguous()
    line = f"{order} rows:"
    for pol in (4, 5, 4, 5):
        t = dga.tiling(m_max, n, k, groups=G, expected_m=m_max)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 128, 256, 2, 2, 3, pol, 0, 1
        f_packed = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((tok.view(G, m_max, k), tsf.view(G, m_max, kb)), (b, sfb), out_packed, mm, m_max, tiling_=t)
        f_index = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(tok, tsf, 0, kb, (b, sfb), out_rows, idx, mm, m_max, m_max, tiling_=t)
        up = min(timeit(f_packed, iters=30, warm=20) for _ in range(3))
        ui = min(timeit(f_index, iters=30, warm=20) for _ in range(3))
        line += f"  p{pol}: packed {up:.1f} indexed {ui:.1f}"
    print(line, flush=True)
