"""Non-temporal B loads in the persistent grouped kernel against the default policy, by rows per expert (development aid;
DGA_B_NT=0/1 is read once per process, so each setting runs in its own process: pass the setting as argv[1])."""
import os, sys
os.environ["DGA_B_NT"] = sys.argv[1]
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from widen_perf import timeit
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(7)
G, n, k = 256, 2048, 7168
kb, nb = k // 128, n // 128
rf = lambda sh: (lambda x: torch.where((x & 0x7F) == 0x7F, x & 0x80, x))(torch.randint(0, 256, sh, dtype=torch.uint8, device=dev, generator=gen))
line = f"DGA_B_NT={sys.argv[1]}:"
for (m_max, hi) in [(128, 128), (128, -112), (128, -96), (128, -64), (64, -64), (64, -48), (64, -32), (16, -16), (16, -8)]:   # hi < 0: every expert has exactly -hi rows
    a = rf((G, m_max, k)); b = rf((G, n, k))
    sfa = torch.rand((G, m_max, kb), device=dev, generator=gen) + 0.5
    sfb = torch.rand((G, nb, kb), device=dev, generator=gen) + 0.5
    o = torch.empty((G, m_max, n), dtype=torch.bfloat16, device=dev)
    mm = torch.randint(0, hi + 1, (G,), dtype=torch.int32, device=dev, generator=gen) if hi > 0 else torch.full((G,), -hi, dtype=torch.int32, device=dev)
    if hi == m_max == 128: mm = torch.full((G,), 128, dtype=torch.int32, device=dev)
    t = dga.tiling(m_max, n, k, groups=G, expected_m=m_max)
    fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o, mm, m_max, tiling_=t)
    line += f"  m_max{m_max}/rows<= {hi}: {min(timeit(fn, iters=30, warm=30) for _ in range(3)):.1f}"
    del a, b, sfa, sfb, o
print(line, flush=True)
