"""One decode shape under the cold-cache protocol (operand sets rotated past the Infinity Cache), the tuned table's tiling:
launch-to-launch time by HIP events, and -- run under `rocprofv3 --kernel-trace --stats` -- the split between the tile kernel
and the split-K combine kernel.  usage: python scripts/decode_breakdown.py M N K [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga

m, n, k = (int(x) for x in sys.argv[1:4])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 120
a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128), n, k, seed=0)
a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
opbytes = m * k + n * k + 2 * m * n
sets = max(3, -(-320 * 2 ** 20 // opbytes))
copies = [(a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(sets)]
t = dga.tiling(m, n, k)
fns = [(lambda c=c: dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=t)) for c in copies]
for i in range(3 * sets):
    fns[i % sets]()
torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(iters):
    fns[i % sets]()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
print(f"{m}x{n}x{k}: tile {t.m1}x{t.n1} stages {t.stages} split-K {t.splitkFactor} policy {t.dispatchPolicyTag}: cold {us:.1f} us per call "
      f"({opbytes / us / 1e3:.0f} GB/s of operands; {sets} operand sets)")
