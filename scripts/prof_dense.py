"""Launch the dense fp8 GEMM a few times (for rocprofv3 runs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

m, n, k = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096)))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
var = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else None
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
t = dga.tiling(m, n, k)
if var:
    t.m1, t.n1 = var
for _ in range(iters):
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
torch.cuda.synchronize()
print("done", m, n, k, t.m1, t.n1)
