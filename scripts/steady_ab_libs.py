"""Builds of libdga_hip.so side by side at sustained clocks (each gets a 300 ms warm run): 4096^3 dense, default tiling.
usage: python scripts/steady_ab_libs.py a.so b.so ...   (outputs of ablation builds are NOT comparable; time only)"""
import ctypes, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import os
m, n, k = (int(v) for v in os.environ.get('DGA_AB_SHAPE', '4096,4096,4096').split(','))
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for path in sys.argv[1:]:
    L = ctypes.CDLL(path)
    L.dga_gemm_fp8_fp8_bf16_nt.restype = ctypes.c_int
    L.dga_gemm_fp8_fp8_bf16_nt.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    fn = lambda: L.dga_gemm_fp8_fp8_bf16_nt(a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), out.data_ptr(), m, n, k, None, None, 0, st)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(50): fn()
        torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 5)
    print(f"{Path(path).name}: {best:.1f} us  {2.0*m*n*k/best/1e6:.0f} TFLOP/s", flush=True)
