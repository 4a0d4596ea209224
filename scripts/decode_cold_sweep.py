"""Decode-shape tilings under the cold-cache protocol: every candidate of harness/sweep.candidates() timed on operand sets
rotated past the Infinity Cache (what a real decode step sees: each layer's weights come from HBM) and, beside it, warm
(one set re-launched).  Prints the best few of each and writes a jsonl (development aid behind tuned/mi355x.csv)."""
import sys, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from deepgemm_ascend_amd.harness import sweep

out_path = Path(sys.argv[1]) if len(sys.argv) > 1 else Path("gpurun_out/decode_cold_sweep.jsonl")
shapes = [(8, 18432, 7168), (8, 7168, 18432), (64, 18432, 7168), (64, 7168, 18432), (64, 4096, 7168), (64, 24576, 1536),
          (64, 32768, 512), (64, 7168, 16384), (128, 18432, 7168), (128, 7168, 18432), (128, 4096, 7168)]
if len(sys.argv) > 2:
    shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2:]]


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


with open(out_path, "w") as f:
    for (m, n, k) in shapes:
        a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128), n, k, seed=0)
        a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
        opbytes = m * k + n * k + 2 * m * n
        sets = max(3, -(-320 * 2 ** 20 // opbytes))
        copies = [(a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty((m, n), dtype=torch.bfloat16, device="cuda")) for _ in range(sets)]
        res = []
        seen = set()
        for c in sweep.candidates(m, n, k, [0]):
            key = (c["m1"], c["n1"], c["stages"], c["splitk"], c["policy"])
            if key in seen or c["policy"] in (1, 2) and c["m1"] != 256:
                continue
            seen.add(key)
            t = dga.tiling(m, n, k)
            t.m1, t.n1, t.swizzleOffset = c["m1"], c["n1"], c["raster"]
            t.stages, t.wavesM, t.wavesN, t.dispatchPolicyTag = c["stages"], 0, 0, c["policy"]
            t.splitkFactor = c["splitk"]; t.kernelSerial = 4 if c["splitk"] > 1 else 0
            try:
                fns = [(lambda cc=cc: dga.gemm_fp8_fp8_bf16_nt((cc[0], cc[1]), (cc[2], cc[3]), cc[4], tiling_=t)) for cc in copies]
                fns[0](); torch.cuda.synchronize()
                cold = rotating(fns, iters=max(24, 3 * sets))
                warm = rotating(fns[:1], iters=40)
            except Exception as e:
                continue
            r = {"shape": [m, n, k], "cold_us": round(cold, 2), "warm_us": round(warm, 2), **c}
            res.append(r); f.write(json.dumps(r) + "\n"); f.flush()
        cur = dga.tiling(m, n, k)
        bc = sorted(res, key=lambda r: r["cold_us"])[:4]
        bw = sorted(res, key=lambda r: r["warm_us"])[:2]
        fmt = lambda r: f"{r['m1']}x{r['n1']} st{r['stages']} sk{r['splitk']} p{r['policy']}: cold {r['cold_us']} warm {r['warm_us']}"
        print(f"{m}x{n}x{k} (table: {cur.m1}x{cur.n1} st{cur.stages} sk{cur.splitkFactor} p{cur.dispatchPolicyTag}; {opbytes / 1e6:.0f} MB x {sets} sets)", flush=True)
        for r in bc: print("   cold-best ", fmt(r), f"-> {opbytes / r['cold_us'] / 1e3:.0f} GB/s", flush=True)
        for r in bw: print("   warm-best ", fmt(r), flush=True)
        del copies
