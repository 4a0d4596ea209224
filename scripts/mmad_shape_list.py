"""The reference's own benchmark, as it runs it: run_mmad_bench(x[M,K] fp16, y[K,N] fp16, z[M,N] f32) over the 18-shape sweep list
(framework/benchmark/benchmark.py:24-44, :364-377).  Device time of the C entry (dga_run_mmad_rtc_ws, batch 1 -- the launch
run_mmad_bench makes after filling its parameter block) by graph replay, with the bound that applies (dense fp16 matrix peak 2.5
PFLOP/s, HBM 8 TB/s: 2(MK + KN) + 4MN bytes) and the fraction reached.  Usage: python scripts/mmad_shape_list.py [--cold]"""
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402
from deepgemm_ascend_amd import _lib, api  # noqa: E402
from deepgemm_ascend_amd.harness import sweep  # noqa: E402


def main():
    cold = "--cold" in sys.argv
    lib = _lib.lib()
    for (m, n, k) in sweep.SHAPE_GROUP:
        g = torch.Generator(device="cuda").manual_seed(m + n + k)
        nset = max(1, min(16, (320 << 20) // (2 * (m * k + n * k) + 1) + 1)) if cold and m <= 256 else 1
        sets = [((torch.randn((m, k), device="cuda", generator=g) * 0.5).to(torch.float16),
                 (torch.randn((k, n), device="cuda", generator=g) * 0.5).to(torch.float16),
                 torch.empty((m, n), dtype=torch.float32, device="cuda")) for _ in range(nset)]
        turn = [0]
        def fn():
            x, y, z = sets[turn[0] % nset]; turn[0] += 1
            ws_ptr, ws_bytes = api._mmad_workspace(1, m, n, k, x)
            rc = lib.dga_run_mmad_rtc_ws(x.data_ptr(), y.data_ptr(), z.data_ptr(), 1, m, n, k, api._dt16(x), ws_ptr, ws_bytes,
                                         api._stream_ptr(z))
            assert rc == 0, rc
        fn(); torch.cuda.synchronize()
        x, y, z = sets[0]
        ref = x.float() @ y.float()
        err = float((z - ref).abs().max() / ref.abs().max())
        n_it = nset * max(1, 16 // nset)
        us = min(u for u in (sweep.graph_us(fn, n_it, replays=3) for _ in range(2)) if u)
        flops, byt = 2.0 * m * n * k, 2.0 * (m * k + n * k) + 4.0 * m * n
        t_m, t_h = flops / 2.5e9, byt / 8e6
        print(json.dumps({"m": m, "n": n, "k": k, "us": round(us, 2), "tflops": round(flops / us / 1e6, 1), "gbps": round(byt / us / 1e3, 1),
                          "bound": "mfma" if t_m >= t_h else "hbm", "frac": round(max(t_m, t_h) / us, 3), "rel_err": round(err, 6),
                          "cold": bool(cold and m <= 256)}), flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
