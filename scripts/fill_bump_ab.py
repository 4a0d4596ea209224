"""The fp8 selector's half-fill split bump (csrc/dga_tiling.cpp: a decode raster that fills at most half of the CUs with the split
the fitted model chose gets its split doubled while a slice keeps >= 16 k blocks) against the model's own split, on shapes off the
tuned table; cold, device time by graph replay.  The model's splits come from a child process with $DGA_NO_FILL_BUMP = 1 (the
switch is read once per process).  Usage: python scripts/fill_bump_ab.py"""
import json
import math
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
SHAPES = [(m, n, k) for n in (11008, 12288, 13312, 15360, 16000) for k in (2048, 4096, 5120, 8192) for m in (32, 64, 128)]
SHAPES += [(m, n, k) for (n, k) in ((6656, 4096), (7680, 6144), (3584, 8192), (2560, 10240)) for m in (48, 96)]
CHILD = '''
import os, sys
sys.path.insert(0, %r)
import deepgemm_ascend_amd as dga
print(" ".join(str(int(dga.select_kernel(*s).splitkFactor)) for s in %r))
''' % (str(ROOT), SHAPES)


def main():
    env = dict(os.environ, DGA_NO_FILL_BUMP="1")
    model = [int(x) for x in subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, check=True).stdout.split()]
    sys.path.insert(0, str(ROOT))
    import torch
    import deepgemm_ascend_amd as dga
    from deepgemm_ascend_amd.harness import sweep
    ratios = []
    for (m, n, k), s_model in zip(SHAPES, model):
        t_b = dga.select_kernel(m, n, k)
        if int(t_b.splitkFactor) == s_model:
            continue
        a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
        nset = min(16, max(3, -(-(320 << 20) // (m * k + n * k + 2 * m * n))))
        sets = [(a, sfa, b, sfb, torch.empty((m, n), dtype=torch.bfloat16, device="cuda"))]
        for _ in range(nset - 1):
            sets.append((a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty((m, n), dtype=torch.bfloat16, device="cuda")))
        n_it = nset * max(1, 16 // nset)
        out = {}
        for name, s in (("bump", int(t_b.splitkFactor)), ("model", s_model)) * 2:
            tt = dga.select_kernel(m, n, k)
            tt.splitkFactor = s; tt.kernelSerial = 4 if s > 1 else 0
            tt.blockDim = -(-m // tt.m1) * -(-n // tt.n1) * s
            turn = [0]

            def fn():
                c = sets[turn[0] % nset]; turn[0] += 1
                dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=tt)
            fn(); torch.cuda.synchronize()
            ok, _ = sweep.is_correct(golden, sets[0][4], s_abs, short_k=k < 128)
            assert ok, (m, n, k, name)
            out[name] = min(out.get(name, 1e30), min(x for x in (sweep.graph_us(fn, n_it, replays=3) for _ in range(2)) if x))
        ratios.append(out["bump"] / out["model"])
        print(json.dumps({"shape": [m, n, k], "tile": f"{t_b.m1}x{t_b.n1}", "model_s": s_model, "bump_s": int(t_b.splitkFactor),
                          "model_us": round(out["model"], 2), "bump_us": round(out["bump"], 2), "ratio": round(ratios[-1], 3)}), flush=True)
        del sets
    print("changed", len(ratios), "of", len(SHAPES), "geomean", round(math.exp(sum(math.log(r) for r in ratios) / max(1, len(ratios))), 4),
          "max", round(max(ratios), 3) if ratios else None)


if __name__ == "__main__":
    main()
