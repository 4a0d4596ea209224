"""The bf16-exact policy on rasters with a partial last round: what the selector picks now against the one-launch Stream-K
(kernelSerial 7 on the 128 x 256 tile) and the persistent kernel, interleaved in one process.
Usage: python scripts/r06_bx_streamk_ab.py [M N K ...]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga

shapes = [(1024, 4096, 7168), (1279, 5003, 7681), (3511, 6151, 8191), (2304, 4096, 7168), (1024, 18432, 7168), (512, 7168, 18432),
          (4096, 2048, 7168), (2048, 7168, 4096), (4096, 4096, 4096)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v), 3)]
for m, n, k in shapes:
    if k % 128 == 0:
        a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    else:   # (the survey's recipe needs whole scale blocks: random e4m3 bytes and scales for the ragged shapes)
        g = torch.Generator(device="cuda").manual_seed(0)
        kb = -(-k // 128)
        a = (torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=g) | (torch.randint(0, 2, (m, k), dtype=torch.uint8, device="cuda", generator=g) << 7))
        b = (torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=g) | (torch.randint(0, 2, (n, k), dtype=torch.uint8, device="cuda", generator=g) << 7))
        sfa = torch.rand((m, kb), device="cuda") + 0.5
        sfb = torch.rand((-(-n // 128), kb), device="cuda") + 0.5
    fns, res, desc = {}, {}, {}
    pick = dga.tiling(m, n, k, policy="bf16_exact")
    desc["pick"] = f"{pick.m1}x{pick.n1} serial {pick.kernelSerial} splitk {pick.splitkFactor}"
    def mk(t):
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        return lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="bf16_exact", tiling_=t)
    fns["pick"] = mk(pick)
    for name, serial, build in (("streamk", 7, 0), ("persistent", 0, 7)):
        t = dga.tiling(m, n, k, policy="bf16_exact")
        t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.stages, t.build, t.wavesM, t.wavesN = 128, 256, serial, 1, 3, build, 0, 0
        fns[name] = mk(t)
    for f in fns.values():
        f()
    torch.cuda.synchronize()
    for rnd in range(3):
        for name, f in fns.items():
            us = bench._graph_us(f, 20) if m * n * k < 2e10 else bench._prewarmed_us(f, 40, 60.0)
            res[name] = min(res.get(name, 1e9), us)
    tiles = -(-m // 128) * -(-n // 256)
    print(f"{m}x{n}x{k} ({tiles} tiles = {tiles / 256:.2f} rounds): pick [{desc['pick']}] {res['pick']:8.2f} us   stream-K {res['streamk']:8.2f} us   "
          f"persistent {res['persistent']:8.2f} us", flush=True)
