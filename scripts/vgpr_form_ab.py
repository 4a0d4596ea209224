import sys; sys.path.insert(0, "/root/repo")
import torch, bench
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
print("fast path: policy 0 (4 computing waves, plain loop) against policy 4 (+ loader waves), three stages")
for (m, n, k, m1, n1, wm, wn) in [(4096, 2048, 7168, 128, 256, 2, 2), (2048, 2048, 7168, 128, 128, 2, 2), (1024, 4096, 7168, 128, 128, 2, 2), (2048, 8192, 2048, 64, 256, 1, 4), (1024, 2048, 4096, 64, 128, 1, 4)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    line = f"{m}x{n}x{k} tile {m1}x{n1}:"
    for pol in (0, 4):
        t = dga.tiling(m, n, k); t.m1, t.n1, t.stages, t.dispatchPolicyTag, t.splitkFactor, t.kernelSerial, t.wavesM, t.wavesN = m1, n1, 3, pol, 1, 0, wm, wn
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
        fn(); torch.cuda.synchronize()
        line += f"  policy {pol}: {sweep.graph_us(fn, 20, 5, 200.0):7.2f} us"
    print(line, flush=True)
print("bf16-exact policy, each build of its menu")
for (m, n, k) in [(1024, 4096, 7168), (2048, 2048, 2048), (512, 4096, 4096)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    line = f"{m}x{n}x{k}:"
    for (m1, n1) in ((128, 256), (128, 128), (64, 256), (64, 128), (32, 128)):
        t = dga.tiling(m, n, k); t.m1, t.n1, t.stages, t.dispatchPolicyTag, t.splitkFactor, t.kernelSerial, t.wavesM, t.wavesN = m1, n1, 3, 7, 1, 0, 0, 0
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
        fn(); torch.cuda.synchronize()
        line += f"  {m1}x{n1}: {sweep.graph_us(fn, 20, 5, 200.0):7.2f}"
    print(line, flush=True)
