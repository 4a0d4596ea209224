"""A/B of two builds of libdga_hip.so on the same device, interleaved rounds (development aid).
usage: python scripts/ab_libs.py a.so b.so [c.so ...]"""
import ctypes, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

libs = {p: ctypes.CDLL(p) for p in sys.argv[1:]}
for L in libs.values():
    L.dga_gemm_fp8_fp8_bf16_nt.restype = ctypes.c_int
    L.dga_gemm_fp8_fp8_bf16_nt.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int] * 3 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (8192, 8192, 8192)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    def run(L):
        rc = L.dga_gemm_fp8_fp8_bf16_nt(a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), out.data_ptr(), m, n, k, None, None, 0, st)
        assert rc == 0, rc
    res = {p: [] for p in libs}
    ref = None
    for pth, L in libs.items():  # every build must give the same bytes
        out.zero_(); run(L); torch.cuda.synchronize()
        if ref is None: ref = out.clone()
        elif not torch.equal(ref.view(torch.int16), out.view(torch.int16)): print(f"  !! {Path(pth).name} differs from the first build", flush=True)
    for rnd in range(7):
        for p, L in libs.items():
            for _ in range(5): run(L)
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run(L)
            e1.record(); torch.cuda.synchronize()
            res[p].append(e0.elapsed_time(e1) * 20)
    for p, v in res.items():
        v = sorted(v)
        print(f"{m}x{n}x{k} {Path(p).name}: median {v[len(v)//2]:.1f} us  min {v[0]:.1f}  ({2.0*m*n*k/v[len(v)//2]/1e6:.0f} TFLOP/s)", flush=True)
