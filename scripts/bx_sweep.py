"""Device-timed sweep of the bf16-exact policy's menu: every (tile, split-K) on a list of shapes, timed by graph replay, short-M
shapes on operand sets rotated past the Infinity Cache -- records in harness/sweep.py's jsonl format (policy 7), the input of
scripts/fit_heuristic.py --bf16-exact.   usage: python scripts/bx_sweep.py OUT_DIR M,N,K [M,N,K ...]"""
import json, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep

out_dir = Path(sys.argv[1]); out_dir.mkdir(parents=True, exist_ok=True)
shapes = [tuple(int(x) for x in s.split(",")) for s in sys.argv[2:]]
MENU = ((128, 256), (128, 128), (64, 256), (64, 128), (32, 128))
for (m, n, k) in shapes:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    opbytes = m * k + n * k + 2 * m * n
    sets = [(a, sfa, b, sfb, out)]
    if m <= sweep.COLD_MAX_M and opbytes < sweep.INFINITY_CACHE:
        for _ in range(min(16, max(3, -(-(320 << 20) // opbytes))) - 1):
            sets.append((a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty_like(out)))
    kb = -(-k // 128)
    recs, idx, turn = [], 0, [0]
    for (m1, n1) in MENU:
        if m1 >= 2 * max(m, 32) and m1 > 32:
            continue
        tiles = -(-m // m1) * -(-n // n1)
        for sk in (1, 2, 3, 4, 5, 6, 8, 16):
            if sk > 1 and (kb // sk < 4 or tiles * sk > 1024 or tiles >= 192):
                continue
            t = dga.tiling(m, n, k)
            t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.dispatchPolicyTag = m1, n1, 3, 0, 0, 7
            t.splitkFactor, t.kernelSerial, t.swizzleOffset = sk, (4 if sk > 1 else 0), sweep.heuristic_raster(m, n, m1, n1, sk, stages=3)

            def fn():
                c = sets[turn[0] % len(sets)]
                turn[0] += 1
                dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=t)
            turn[0] = 0
            fn(); torch.cuda.synchronize()
            ok, diff = sweep.is_correct(golden, out, s_abs, policy="bf16_exact", short_k=k < 128)
            n_it = -(-max(12, 2 * len(sets)) // len(sets)) * len(sets)
            turn[0] = 0
            us = sweep.time_us(fn, warm=max(3, len(sets)), iters=n_it, device_time=True) if ok else -1
            p = {"m1": m1, "n1": n1, "raster": int(t.swizzleOffset), "stages": 3, "splitk": sk, "policy": 7}
            if len(sets) > 1:
                p["cold_sets"] = len(sets)
            recs.append({"idx": idx, "M": m, "N": n, "K": k, "time": us, "diff": diff, "negative": not ok, "parameters": p})
            idx += 1
    (out_dir / f"shape_{m}_{n}_{k}_rank_0.jsonl").write_text("".join(json.dumps(r) + "\n" for r in recs))
    good = [r for r in recs if not r["negative"]]
    if good:
        best = min(good, key=lambda r: r["time"])
        print(m, n, k, "best", best["parameters"]["m1"], best["parameters"]["n1"], best["parameters"]["splitk"], round(best["time"], 2), flush=True)
