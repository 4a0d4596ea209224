import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, str(ROOT))
    import torch, deepgemm_ascend_amd as dga, bench
    for (m, n, k) in [(1024, 18432, 7168), (2048, 5120, 13824), (2048, 4096, 7168)]:
        a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=2)
        o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        tk = dga.tiling(m, n, k)
        tk.m1, tk.n1, tk.wavesM, tk.wavesN, tk.stages, tk.dispatchPolicyTag, tk.kernelSerial, tk.splitkFactor = 256, 256, 4, 2, 2, 2, 7, 1
        fk = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=tk)
        fk(); torch.cuda.synchronize()
        print(f"DGA_SK_DEBUG={os.environ.get('DGA_SK_DEBUG')} {m}x{n}x{k}: {min(bench._prewarmed_us(fk, 60, 100.0) for _ in range(2)):.1f} us", flush=True)
else:
    for v in ("0", "1", "2", "4", "6", "7"):
        subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, DGA_SK_DEBUG=v), check=True)
