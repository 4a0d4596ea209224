import sys
sys.path.insert(0, "/root/repo")
import torch
import deepgemm_ascend_amd as dga
import bench
sys.path.insert(0, "/root/repo/scripts")
from widen_perf import timeit
for (m, n, k, bm, sk) in [(8, 18432, 7168, 16, 1), (8, 18432, 7168, 16, 2), (8, 7168, 18432, 16, 4), (64, 18432, 7168, 64, 1), (64, 7168, 18432, 64, 4), (64, 4096, 7168, 64, 8)]:
    a0, sfa0, b, sfb = bench.make_dense_inputs(128, n, k, seed=0)
    a, sfa = a0[:m].contiguous(), sfa0[:m].contiguous()
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    line = f"{m}x{n}x{k} {bm}x128 st3 sk{sk}:"
    outs = []
    for pol in (0, 4):
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.stages, t.splitkFactor, t.wavesM, t.wavesN, t.dispatchPolicyTag, t.kernelSerial = bm, 128, 3, sk, 0, 0, pol, (4 if sk > 1 else 0)
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
        us = min(timeit(fn, iters=100, warm=50) for _ in range(3))
        torch.cuda.synchronize(); outs.append(out.clone())
        line += f" policy {pol}: {us:.1f} us |"
    print(line, "same bits:", torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), flush=True)
