"""Host time per operator call (Python -> C ABI -> hipLaunchKernel returns) against the device time of the same call, for the
short shapes of the reference's sweep list: is a back-to-back stream of calls bound by the host?"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
try:
    from deepgemm_ascend_amd import deep_gemm_cpp
except Exception as e:
    deep_gemm_cpp = None
    print("deep_gemm_cpp:", repr(e))
for (m, n, k) in [(64, 32768, 512), (64, 24576, 1536), (64, 4096, 7168), (128, 4096, 7168), (4096, 4096, 4096)]:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    variants = {"api, tiling looked up per call": lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out),
                "api, tiling passed": lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)}
    if deep_gemm_cpp is not None:
        a8, b8 = a.view(torch.float8_e4m3fn), b.view(torch.float8_e4m3fn)
        variants["deep_gemm_cpp (pybind)"] = lambda: deep_gemm_cpp.gemm_fp8_fp8_bf16_nt(a8, sfa, b8, sfb, out)
    for name, fn in variants.items():
        for _ in range(50): fn()
        torch.cuda.synchronize()
        N = 2000
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter(); e0.record()
        for _ in range(N): fn()
        host = (time.perf_counter() - t0) / N * 1e6
        e1.record(); torch.cuda.synchronize()
        dev = e0.elapsed_time(e1) * 1e3 / N
        print(f"{m}x{n}x{k} sk{t.splitkFactor} {name:34s}: host {host:6.2f} us/call issued, stream {dev:6.2f} us/call", flush=True)

import cProfile, pstats
m, n, k = 64, 32768, 512
a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
pr = cProfile.Profile()
pr.enable()
for _ in range(3000):
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
