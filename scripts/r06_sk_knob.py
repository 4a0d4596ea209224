import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
import deepgemm_ascend_amd as dga
for m, n, k in ((1279, 5003, 7680), (2048, 7168, 4096), (3511, 6151, 8192)):
    g = torch.Generator(device="cuda").manual_seed(0)
    kb = -(-k // 128)
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=g)
    sfa = torch.rand((m, kb), device="cuda") + 0.5
    sfb = torch.rand((-(-n // 128), kb), device="cuda") + 0.5
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    row = []
    for knob in (0, 2, 4, 5):
        os.environ["DGA_BXSK_KNOB"] = str(knob)
        t = dga.tiling(m, n, k, policy="bf16_exact")
        t.m1, t.n1, t.kernelSerial, t.splitkFactor, t.stages, t.build, t.wavesM, t.wavesN = 128, 256, 7, 1, 3, 0, 0, 0
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="bf16_exact", tiling_=t)
        fn(); torch.cuda.synchronize()
        us = min(bench._prewarmed_us(fn, 40, 60.0) for _ in range(2))
        row.append(f"knob {knob}: {us:7.1f}")
    t.kernelSerial, t.build = 0, 7
    us = min(bench._prewarmed_us(fn, 40, 60.0) for _ in range(2))
    print(f"{m}x{n}x{k}: " + "  ".join(row) + f"  persistent {us:7.1f}  (knob 2: tails only, 4: mains do not add, 5: mains alone)", flush=True)
