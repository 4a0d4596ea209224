"""Decode shapes cold (operand sets rotated past the Infinity Cache): the tuned tiling against deeper LDS rings (stages 4..6) of the
same tile, and neighbouring split-K factors (profiles/r03_deep_ring_cold.txt; the 4..6-stage builds existed only for that
measurement: today those rows report DGA_E_TILING).  usage: python scripts/deep_ring_cold.py [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import copy
import torch
import bench
import deepgemm_ascend_amd as dga

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120


def cold_us(m, n, k, t, copies):
    sets = len(copies)
    fns = [(lambda c=c: dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=t)) for c in copies]
    for i in range(2 * sets):
        fns[i % sets]()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        fns[i % sets]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


SHAPES = [(8, 18432, 7168), (8, 7168, 18432), (64, 4096, 7168), (64, 7168, 18432), (64, 18432, 7168), (128, 4096, 7168), (128, 7168, 18432),
          (128, 18432, 7168), (64, 24576, 1536)]
CANDS = {  # (m1, n1) -> stages to try
    (64, 128): (3, 4, 5, 6), (32, 128): (3, 4, 6), (16, 128): (3, 4, 6), (16, 256): (3, 4), (32, 256): (3, 4), (64, 256): (3,), (128, 256): (3,),
}
for (m, n, k) in SHAPES:
    a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128), n, k, seed=0)
    a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
    opbytes = m * k + n * k + 2 * m * n
    sets = max(3, -(-320 * 2 ** 20 // opbytes))
    copies = [(a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(sets)]
    t0 = dga.tiling(m, n, k)
    base = cold_us(m, n, k, t0, copies)
    print(f"{m}x{n}x{k}: tuned {t0.m1}x{t0.n1} st{t0.stages} sk{t0.splitkFactor} pol{t0.dispatchPolicyTag}: {base:.1f} us ({opbytes / base / 1e3:.0f} GB/s)", flush=True)
    ref = None
    tiles = [(t0.m1, t0.n1)] + [x for x in ((64, 128), (32, 128), (16, 256), (32, 256), (16, 128)) if x != (t0.m1, t0.n1) and x[0] >= min(m, 16) and x[0] <= max(16, m)]
    for (m1, n1) in tiles:
        for st in CANDS.get((m1, n1), (3,)):
            for sk in sorted({max(1, t0.splitkFactor // 2), t0.splitkFactor, t0.splitkFactor * 2 if t0.splitkFactor * 2 <= 16 else t0.splitkFactor}):
                for pol in (0, 4):
                    t = copy.copy(t0) if hasattr(t0, "__copy__") else dga.tiling(m, n, k)
                    t = dga.tiling(m, n, k)
                    t.m1, t.n1, t.stages, t.splitkFactor, t.dispatchPolicyTag = m1, n1, st, sk, pol
                    t.wavesM, t.wavesN = 0, 0
                    t.kernelSerial = 4 if sk > 1 else 0
                    try:
                        us = cold_us(m, n, k, t, copies)
                    except Exception as e:
                        print(f"   {m1}x{n1} st{st} sk{sk} pol{pol}: {e!r}"[:160], flush=True)
                        continue
                    if ref is None:
                        out0 = copies[0][4].clone()
                    mark = " <-- best so far" if us < base * 0.97 else ""
                    print(f"   {m1:3d}x{n1:3d} st{st} sk{sk:2d} pol{pol}: {us:6.1f} us ({us / base:.2f}x){mark}", flush=True)
    del copies
