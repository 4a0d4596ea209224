"""Launch the grouped masked-M GEMM (BASELINE configs[3]: 256 x (M <= 128, K = 7168, N = 2048)) a few times, for rocprofv3 runs.
usage: prof_grouped.py [launches] [full|random] [policy]   -- `random`: masked_m ~ randint(0, 129), SURVEY.md 8(d)'s second mask;
policy: bf16_exact (the operator's default) | fast | strict"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
G, MMAX, N, K = 256, 128, 2048, 7168
g = torch.Generator(device="cuda").manual_seed(0)
def rf(shape):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=g)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)
a = rf((G, MMAX, K)); b = rf((G, N, K))
sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5; sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
mask = sys.argv[2] if len(sys.argv) > 2 else "full"
masked = (torch.full((G,), MMAX, dtype=torch.int32, device="cuda") if mask == "full" else
          torch.randint(0, MMAX + 1, (G,), dtype=torch.int32, device="cuda", generator=g))
policy = sys.argv[3] if len(sys.argv) > 3 else None
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, policy=policy)
torch.cuda.synchronize()
print("done")
