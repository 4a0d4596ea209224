"""Decode shapes (the reference's benchmark list is mostly M = 8..128): two against three LDS stages and split-K factors,
warm (one operand set) and cold (operand sets rotated past the Infinity Cache).  Development aid."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

def timeit(fns, iters=120, warm=30):
    n = len(fns)
    best = 1e30
    for _ in range(3):
        for i in range(warm): fns[i % n]()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fns[i % n]()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

for (m, n, k) in [(8, 18432, 7168), (8, 7168, 18432), (64, 18432, 7168), (64, 7168, 18432), (64, 4096, 7168), (128, 4096, 7168), (128, 18432, 7168), (128, 7168, 18432)]:
    a0, sfa0, b0, sfb0 = bench.make_dense_inputs(128, n, k, seed=0)
    sets = max(3, -(-320 * 2 ** 20 // (n * k)))
    ops = [(a0[:m].clone(), sfa0[:m].clone(), b0.clone(), sfb0.clone(), torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(sets)]
    t0 = dga.tiling(m, n, k)
    bm = 16 if m <= 16 else 32 if m <= 32 else 64 if m <= 64 else 128
    line = f"{m}x{n}x{k} (auto {t0.m1}x{t0.n1} st{t0.stages} sk{t0.splitkFactor} p{t0.dispatchPolicyTag}):"
    cands = {"auto": None}
    for bn in (128, 256):
        for st in (2, 3):
            for sk in (1, 2, 4, 8):
                if bm == 128 and st == 2 and bn == 256: continue
                cands[f"{bm}x{bn} st{st} sk{sk}"] = (bm, bn, st, sk)
    res = {}
    for name, c in cands.items():
        t = dga.tiling(m, n, k)
        if c:
            kb = k // 128
            if c[3] > 1 and kb // c[3] < 4: continue
            t.m1, t.n1, t.stages, t.splitkFactor = c
            t.wavesM = t.wavesN = 0
            t.dispatchPolicyTag = 4 if (c[2] == 3 and (c[0], c[1]) in ((128, 256), (128, 128), (64, 256))) else 0
            t.kernelSerial = 4 if c[3] > 1 else 0
        try:
            fns = [(lambda o=o, t=t: dga.gemm_fp8_fp8_bf16_nt((o[0], o[1]), (o[2], o[3]), o[4], tiling_=t)) for o in ops]
            res[name] = (timeit(fns[:1]), timeit(fns))
        except Exception as e:
            res[name] = None
    ok = {k_: v for k_, v in res.items() if v}
    bw = min(ok, key=lambda x: ok[x][0]); bc = min(ok, key=lambda x: ok[x][1])
    print(line, f"auto warm {ok['auto'][0]:.1f} cold {ok['auto'][1]:.1f} | best warm: {bw} {ok[bw][0]:.1f} (cold {ok[bw][1]:.1f}) | best cold: {bc} {ok[bc][1]:.1f} (warm {ok[bc][0]:.1f})", flush=True)
    del ops
