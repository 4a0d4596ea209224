"""Non-temporal output stores / weight loads on the contiguous layout with one 128-row block per group (development aid;
DGA_OUT_NT and DGA_B_NT from argv, read once per process)."""
import os, sys
os.environ["DGA_OUT_NT"] = sys.argv[1]
os.environ["DGA_B_NT"] = sys.argv[2]
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from widen_perf import timeit
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(7)
rf = lambda sh: (lambda x: torch.where((x & 0x7F) == 0x7F, x & 0x80, x))(torch.randint(0, 256, sh, dtype=torch.uint8, device=dev, generator=gen))
line = f"OUT_NT={sys.argv[1]} B_NT={sys.argv[2]}:"
for (groups, rows, n, k) in [(256, 128, 2048, 7168), (32, 128, 4096, 7168), (256, 128, 7168, 2048)]:
    m = groups * rows
    kb, nb = k // 128, n // 128
    a, b = rf((m, k)), rf((groups, n, k))
    sfa = torch.rand((m, kb), device=dev, generator=gen) + 0.5
    sfb = torch.rand((groups, nb, kb), device=dev, generator=gen) + 0.5
    idx = torch.arange(groups, device=dev, dtype=torch.int32).repeat_interleave(rows).contiguous()
    o = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    t = dga.tiling(m, n, k, groups=groups, contiguous=True)
    fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), o, idx, tiling_=t)
    line += f"  {groups}x{rows} n{n} k{k} ({t.m1}x{t.n1} p{t.dispatchPolicyTag}): {min(timeit(fn, iters=30, warm=30) for _ in range(3)):.1f}"
    del a, b, sfa, sfb, o
print(line, flush=True)
