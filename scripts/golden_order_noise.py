"""How far an fp32 matmul golden of unspecified summation order (torch.matmul / np.matmul on the dequantised operands: the
reference harness's golden, gen_golden.py:17-22) is from the oracle-order result: the strict kernel (bit-identical to the CPU
oracle) and the bf16-exact kernel checked against it with the product's bar."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep, tolerance
for (m, n, k) in [(128, 128, 128), (256, 512, 1024), (240, 2688, 12032), (1024, 1024, 4096), (2048, 2048, 7168), (512, 4096, 16384), (4096, 4096, 4096)]:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    want = tolerance.bf16_round(golden)
    line = f"{m}x{n}x{k}:"
    for pol in ("strict", "bf16_exact", "fast"):
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy=None if pol == "fast" else pol, sync=True)
        ok, rep = tolerance.check(out.float(), want, s_abs, policy="fast")
        line += f"  {pol}: beyond 2 ulp {rep['frac_gt_2ulp']:.2e}, worst excess 2^{__import__('math').log2(max(rep['worst_excess_over_S'], 1e-300)):.1f} S, max {rep['max_ulp']:.0f} ulp |"
    print(line, flush=True)
