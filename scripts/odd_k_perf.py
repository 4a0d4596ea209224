"""Odd-K shapes of the reference's sweep list (benchmark.py:41-43): whole call (re-layout pass + tile kernel) and the re-layout pass
alone (a call on the same operands with K rounded down to a multiple of 128 is the tile kernel's share)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
for (m, n, k) in [(1279, 5003, 7681), (3511, 6151, 8191), (5119, 6997, 9901)]:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)
    fn(); torch.cuda.synchronize()
    ok, diff = sweep.is_correct(golden, out, s_abs)
    us = sweep.time_us(fn, warm=5, iters=30)
    kp = -(-k // 128) * 128
    a2, sfa2, b2, sfb2, _, _ = sweep.gen_data(m, n, kp)
    t2 = dga.tiling(m, n, kp); t2.m1, t2.n1, t2.stages, t2.dispatchPolicyTag, t2.splitkFactor, t2.kernelSerial = t.m1, t.n1, t.stages, t.dispatchPolicyTag, t.splitkFactor, t.kernelSerial
    t2.wavesM, t2.wavesN = t.wavesM, t.wavesN
    fn2 = lambda: dga.gemm_fp8_fp8_bf16_nt((a2, sfa2), (b2, sfb2), out, tiling_=t2)
    us2 = sweep.time_us(fn2, warm=5, iters=30)
    print(f"{m} x {n} x {k}: tile {t.m1}x{t.n1} policy {t.dispatchPolicyTag}: call {us:.1f} us ({2.0*m*n*k/us/1e6:.0f} TF), same tiling at K={kp} without the pass "
          f"{us2:.1f} us -> pass {us-us2:.1f} us for {(m+n)*(k+kp)/1e6:.1f} MB = {(m+n)*(k+kp)/(us-us2)/1e6:.2f} TB/s  ok={ok}", flush=True)
# 16-bit path (run_mmad_bench layout needs the pass for x when K % 64 != 0)
for (m, n, k) in [(1279, 5003, 7681)]:
    x = torch.randn(m, k, device="cuda", dtype=torch.float16); y = torch.randn(n, k, device="cuda", dtype=torch.float16)
    o = torch.empty(m, n, device="cuda", dtype=torch.float16)
    try:
        fn = lambda: dga.catlass_dynamic_matmul(x, y.t(), o)
        fn(); torch.cuda.synchronize()
        us = sweep.time_us(fn, warm=3, iters=20)
        ref = (x.float() @ y.float().t())
        err = ((o.float() - ref).abs().max() / ref.abs().max()).item()
        print(f"fp16 op {m} x {n} x {k}: {us:.1f} us ({2.0*m*n*k/us/1e6:.0f} TF) rel err {err:.2e}")
    except Exception as e:
        print("fp16 op:", repr(e))
