"""A/B of the hardware-scale builds (policy "fast_ue8m0": block scales in the MFMA's E8M0 operands, accumulate in place) against
the promotion builds ("fast") on power-of-two scales: same outputs?  parity against the strict kernel, time per launch.
  python scripts/ue8m0_ab.py [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def key(t):
    v = t.view(torch.int16).to(torch.int32)
    mag = v & 0x7FFF
    return torch.where(v < 0, -mag, mag)


def timed(fn, prewarm_ms=300.0):
    return bench._prewarmed_us(fn, iters, prewarm_ms)


print(f"{'shape':>20} {'tile':>9} {'pol':>3} | {'fast us':>8} {'ue8m0 us':>8} {'ratio':>6} | {'TF fast':>8} {'TF ue8m0':>8} | same bf16 | ue8m0 vs strict: max_ulp frac>2ulp")
for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (8192, 8192, 8192), (2048, 4096, 7168), (1024, 4096, 7168), (4096, 7168, 2048),
                  (4096, 4096, 4224), (8192, 4096, 4096), (2048, 5120, 13824)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, (k // 128) * 128, seed=3, ue8m0=True)
    kk = a.shape[1]
    o_f = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    o_u = torch.empty_like(o_f); o_s = torch.empty_like(o_f)
    t = dga.tiling(m, n, kk)
    f_fast = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_f, policy="fast")
    f_ue = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_u, policy="fast_ue8m0")
    f_fast(); f_ue()
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_s, strict=True, sync=True)
    same = float((o_f.view(torch.int16) == o_u.view(torch.int16)).double().mean())
    ul = (key(o_u) - key(o_s)).abs()
    us_f, us_u = timed(f_fast), timed(f_ue)
    fl = 2.0 * m * n * kk
    w4 = ""
    if (t.m1, t.n1) == (256, 256):     # the same tile on FOUR waves (wave tile 128 x 128, accumulators in AGPRs)
        t4 = dga.tiling(m, n, kk)
        t4.wavesM, t4.wavesN, t4.dispatchPolicyTag = 2, 2, 2 | 16
        o_4 = torch.empty_like(o_u)
        f_4 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_4, tiling_=t4)
        f_4(); torch.cuda.synchronize()
        same4 = float((o_4.view(torch.int16) == o_u.view(torch.int16)).double().mean())
        us_4 = timed(f_4)
        w4 = f" | w4: {us_4:8.2f} us {fl / us_4 / 1e6:8.1f} TF same {same4:.7f}"
        del o_4
    print(f"{m:>6}x{n:>6}x{kk:>6} {t.m1:>4}x{t.n1:<4} {t.dispatchPolicyTag:>3} | {us_f:8.2f} {us_u:8.2f} {us_u / us_f:6.3f} | {fl / us_f / 1e6:8.1f} {fl / us_u / 1e6:8.1f} | "
          f"{same:9.7f} | {int(ul.max())} {float((ul > 2).double().mean()):.3e}{w4}", flush=True)
    del a, b, o_f, o_u, o_s

# grouped masked stream (BASELINE configs[3])
from deepgemm_ascend_amd import parallel
g, mmax, n, k = 256, 128, 2048, 7168
gen = torch.Generator(device="cuda").manual_seed(5)
xb = torch.randn((g, n, k), device="cuda", generator=gen)
sb = torch.exp2(torch.ceil(torch.log2(xb.view(g, n // 128, 128, k // 128, 128).abs().amax(dim=(2, 4)).clamp_min(1e-30) / 448.0)))
qb = (xb.view(g, n // 128, 128, k // 128, 128) / sb[:, :, None, :, None]).reshape(g, n, k).to(torch.float8_e4m3fn).view(torch.uint8)
del xb
xa = torch.randn((g, mmax, k), device="cuda", generator=gen)
sa = torch.exp2(torch.ceil(torch.log2(xa.view(g, mmax, k // 128, 128).abs().amax(dim=3).clamp_min(1e-30) / 448.0)))
qa = (xa.view(g, mmax, k // 128, 128) / sa[..., None]).reshape(g, mmax, k).to(torch.float8_e4m3fn).view(torch.uint8)
del xa
for name, mask in (("full", torch.full((g,), mmax, dtype=torch.int32, device="cuda")),
                   ("random", torch.randint(0, mmax + 1, (g,), generator=torch.Generator().manual_seed(99)).to(torch.int32).cuda())):
    outs = {}
    for pol in ("fast", "fast_ue8m0"):
        o = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((qa, sa), (qb, sb), o, mask, mmax, policy=pol)
        fn(); torch.cuda.synchronize()
        us = bench._prewarmed_us(fn, 20, 100.0)
        outs[pol] = (o, us)
    same = float((outs["fast"][0].view(torch.int16) == outs["fast_ue8m0"][0].view(torch.int16)).double().mean())
    print(f"grouped 256 x (128, 7168, 2048) mask {name}: fast {outs['fast'][1]:.1f} us, fast_ue8m0 {outs['fast_ue8m0'][1]:.1f} us, same bf16 {same:.7f}", flush=True)
