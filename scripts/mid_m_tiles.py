"""Mid-M shapes over tile / wave-layout / policy / split-K candidates (profiles/r03_mid_m_tiles.txt).  The 128x128 builds with eight
computing waves (w2x4, w4x2) existed only for that measurement: today those two rows report DGA_E_TILING."""
import sys; sys.path.insert(0, '/root/repo')
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
for (m, n, k) in [(1024, 4096, 7168), (2048, 4096, 7168), (1024, 18432, 7168)]:
    a, sfa, b, sfb, golden, s_abs = sweep.gen_data(m, n, k)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t0 = dga.tiling(m, n, k)
    cands = [("auto", None)]
    for (m1, n1, wm, wn, st, pol, sk) in [(128,128,2,2,3,4,1),(128,128,2,2,3,0,1),(128,128,2,4,3,0,1),(128,128,4,2,3,0,1),(128,128,2,2,3,5,1),
                                           (128,256,2,2,3,4,2),(128,256,2,4,3,0,2),(128,256,2,2,3,4,1),(128,256,2,4,3,0,1),(256,256,4,2,2,2,1),(256,256,4,2,2,2,2),(256,256,4,2,2,2,4)]:
        t = dga.tiling(m, n, k)
        t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.splitkFactor = m1, n1, wm, wn, st, pol, sk
        t.kernelSerial = 4 if sk > 1 else 0
        cands.append((f"{m1}x{n1} w{wm}x{wn} st{st} pol{pol} sk{sk}", t))
    for name, t in cands:
        tt = t if t is not None else t0
        try:
            fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=tt)
            fn(); torch.cuda.synchronize()
            ok, _ = sweep.is_correct(golden, out, s_abs)
            us = sweep.time_us(fn, warm=20, iters=100)
            print(f"{m}x{n}x{k} {name:36s} {us:8.1f} us {2.0*m*n*k/us/1e6:7.0f} TF ok={ok}", flush=True)
        except Exception as e:
            print(f"{m}x{n}x{k} {name}: {e!r}"[:200], flush=True)
