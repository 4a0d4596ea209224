"""Launch the Stream-K kernel and the selector's pick a few times each (for rocprofv3 --kernel-trace --stats)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
m, n, k = (int(x) for x in sys.argv[1:4])
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=2, ue8m0=True)
o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
tk = dga.tiling(m, n, k)
tk.m1, tk.n1, tk.wavesM, tk.wavesN, tk.stages, tk.dispatchPolicyTag, tk.kernelSerial, tk.splitkFactor = 256, 256, 4, 2, 2, 2, 7, 1
for _ in range(60):
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="fast")
for _ in range(60):
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=tk)
torch.cuda.synchronize()
