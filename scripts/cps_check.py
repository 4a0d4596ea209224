"""Persistent continuous-pipeline build (dispatchPolicyTag 6) against the one-tile continuous build (2): byte equality, then
timing on dense shapes of one to four 256x256 tiles per CU (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(3)
ok_all = True
for (m, n, k) in [(256, 256, 256), (512, 768, 384), (2048, 2048, 640), (4096, 4096, 4096), (8192, 4096, 1024), (1024, 18432, 1280),
                  (4096, 7168, 2048), (5120, 4864, 896), (256, 512, 128), (300, 512, 256)]:
    kb, nb = -(-k // 128), -(-n // 128)
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device=dev, generator=gen) | (torch.randint(0, 2, (m, k), dtype=torch.uint8, device=dev, generator=gen) << 7)
    b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device=dev, generator=gen)
    sfa = torch.rand((m, kb), device=dev, generator=gen) + 0.5
    sfb = torch.rand((nb, kb), device=dev, generator=gen) + 0.5
    outs = {}
    for pol in (2, 6):
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 256, 256, 4, 2, 2, pol, 0, 1
        o = torch.full((m, n), -1.0, dtype=torch.bfloat16, device=dev)
        for _ in range(2):
            dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t, sync=True)
        outs[pol] = o
    same = torch.equal(outs[2].view(torch.int16), outs[6].view(torch.int16))
    ok_all &= same
    print(m, n, k, "policy 6 == 2:", same, flush=True)
print("ALL EQUAL" if ok_all else "FAILED", flush=True)
if not ok_all:
    sys.exit(1)
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (8192, 4096, 4096), (4096, 7168, 2048), (2048, 7168, 4096), (1024, 18432, 7168),
                  (4096, 4096, 7168), (6144, 4096, 4096), (4096, 8192, 2048)]:
    a2, sfa2, b2, sfb2 = bench.make_dense_inputs(m, n, k, seed=0)
    o2 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    line = f"dense {m}x{n}x{k} ({(m // 256) * (n // 256)} tiles):"
    for pol in (2, 6, 2, 6):
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 256, 256, 4, 2, 2, pol, 0, 1
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a2, sfa2), (b2, sfb2), o2, tiling_=t)
        line += f"  p{pol} {min(timeit(fn, iters=40 if m * n * k < 2 ** 38 else 15, warm=60) for _ in range(3)):.1f}"
    tt = dga.tiling(m, n, k)
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a2, sfa2), (b2, sfb2), o2, tiling_=tt)
    line += f"  | table/heuristic ({tt.m1}x{tt.n1} p{tt.dispatchPolicyTag} ks{tt.kernelSerial}) {min(timeit(fn, iters=40 if m * n * k < 2 ** 38 else 15, warm=60) for _ in range(3)):.1f}"
    print(line, flush=True)
