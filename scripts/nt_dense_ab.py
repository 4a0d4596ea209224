"""Non-temporal B loads on dense short-M shapes (one tile row: every B panel is read once by one CU), cold protocol, the
persistent loader-wave build forced (dispatchPolicyTag 5).  DGA_B_NT from argv[1] (read once per process)."""
import os, sys
os.environ["DGA_B_NT"] = sys.argv[1]
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

def rotating(fns, iters, warm=12, reps=3):
    n = len(fns)
    for i in range(warm): fns[i % n]()
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fns[i % n]()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

line = f"DGA_B_NT={sys.argv[1]}:"
for (m, n, k, tile) in [(8, 18432, 7168, (16, 128, 1, 4)), (64, 18432, 7168, (64, 128, 1, 4)), (64, 18432, 7168, (64, 256, 1, 4)),
                        (128, 18432, 7168, (128, 128, 2, 2)), (128, 18432, 7168, (128, 256, 2, 2)), (64, 24576, 1536, (64, 128, 1, 4)),
                        (64, 7168, 18432, (64, 128, 1, 4))]:
    a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128), n, k, seed=0)
    a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
    opbytes = m * k + n * k + 2 * m * n
    sets = max(3, -(-(320 << 20) // opbytes))
    copies = [(a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(sets)]
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = *tile, 3, 5, 0, 1
    fns = [(lambda c=c: dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=t)) for c in copies]
    cold = rotating(fns, iters=max(24, 3 * sets)); warm = rotating(fns[:1], iters=40)
    line += f"  {m}x{n}x{k} {tile[0]}x{tile[1]}: cold {cold:.1f} warm {warm:.1f}"
    del copies
print(line, flush=True)
