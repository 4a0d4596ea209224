"""The aclnn operator's own dtypes (fp16 in / out) on the reference's sweep shape list, device time by graph replay, beside the fp8 path's
time on the same shape (a 16-bit operand stream is twice the bytes and the bf16/f16 matrix rate half the fp8 one: ~2x is par)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
for (m, n, k) in sweep.SHAPE_GROUP:
    x = torch.randn(m, k, device="cuda", dtype=torch.float16); y = torch.randn(n, k, device="cuda", dtype=torch.float16)
    o = torch.empty(m, n, device="cuda", dtype=torch.float16)
    fn = lambda: dga.catlass_dynamic_matmul(x, y.t(), o)
    fn(); torch.cuda.synchronize()
    us16 = sweep.graph_us(fn, 10, 3, 20.0)
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    f8 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)
    f8(); torch.cuda.synchronize()
    us8 = sweep.graph_us(f8, 10, 3, 20.0)
    byt = 2 * (m * k + n * k + m * n)
    print(f"{m:5d} x {n:5d} x {k:5d}: fp16 op {us16:8.1f} us ({2.0*m*n*k/us16/1e6:7.1f} TF, {byt/us16/1e3:6.0f} GB/s)   fp8 {us8:8.1f} us   ratio {us16/us8:.2f}", flush=True)
