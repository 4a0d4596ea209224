"""Launch the dense fp8 GEMM a few times (for rocprofv3 runs).
  prof_dense.py [m n k [iters [bm bn]]] [--policy bf16_exact|fast|strict]      (default policy: the operator's own default)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

argv = sys.argv[1:]
policy = None
if "--policy" in argv:
    i = argv.index("--policy")
    policy = argv[i + 1]
    del argv[i:i + 2]
m, n, k = (int(x) for x in (argv[0:3] if len(argv) > 2 else (4096, 4096, 4096)))
iters = int(argv[3]) if len(argv) > 3 else 20
var = (int(argv[4]), int(argv[5])) if len(argv) > 5 else None
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
if var:
    t = dga.tiling(m, n, k, policy=policy if policy == "bf16_exact" else None)
    t.m1, t.n1 = var
    for _ in range(iters):
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t, policy=policy)
else:
    for _ in range(iters):
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy=policy)
torch.cuda.synchronize()
print("done", m, n, k, policy, var)
