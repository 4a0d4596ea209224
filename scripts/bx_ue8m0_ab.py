"""The in-contract arithmetic on power-of-two scales: policy "bf16_exact_ue8m0" (scales folded into the e4m3 -> bf16 conversions,
the bf16 MFMA accumulates in place; four waves, AGPR accumulators) against "bf16_exact": time, parity against the strict kernel."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def key(t):
    v = t.view(torch.int16).to(torch.int32)
    mag = v & 0x7FFF
    return torch.where(v < 0, -mag, mag)


print(f"{'shape':>20} | {'bf16_exact':>10} {'tile':>8} | {'bx_ue8m0':>9} {'ratio':>6} {'TF':>7} | vs strict: max_ulp frac>2ulp | {'fast_ue8m0':>10}")
for (m, n, k) in [(4096, 4096, 4096), (4096, 2048, 7168), (8192, 8192, 8192), (2048, 4096, 7168), (1024, 4096, 7168), (4096, 7168, 2048), (512, 4096, 7168)]:
    a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=3, ue8m0=True)
    o = torch.empty((m, n), dtype=torch.bfloat16, device="cuda"); o_u = torch.empty_like(o); o_s = torch.empty_like(o)
    t = dga.tiling(m, n, k, policy="bf16_exact")
    f0 = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="bf16_exact")
    fu = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_u, policy="bf16_exact_ue8m0")
    ff = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, policy="fast_ue8m0")
    f0(); fu()
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o_s, strict=True, sync=True)
    ul = (key(o_u) - key(o_s)).abs()
    us0 = min(bench._prewarmed_us(f0, iters, 200.0) for _ in range(2))
    usu = min(bench._prewarmed_us(fu, iters, 200.0) for _ in range(2))
    usf = min(bench._prewarmed_us(ff, iters, 200.0) for _ in range(2))
    print(f"{m:>6}x{n:>6}x{k:>6} | {us0:10.2f} {t.m1:>4}x{t.n1:<3} | {usu:9.2f} {usu / us0:6.3f} {2.0 * m * n * k / usu / 1e6:7.1f} | {int(ul.max()):>7} {float((ul > 2).double().mean()):.3e} | {usf:10.2f}", flush=True)
    del a, b, o, o_u, o_s

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
alg = g * n * k + g * mmax * (k + 4 * (k // 128) + 2 * n)
for name, mask, em in (("full", torch.full((g,), mmax, dtype=torch.int32, device="cuda"), 128),
                       ("random", torch.randint(0, mmax + 1, (g,), generator=torch.Generator().manual_seed(99)).to(torch.int32).cuda(), 128)):
    res = {}
    for pol in ("fast", "bf16_exact", "bf16_exact_ue8m0"):
        o = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((qa, sa), (qb, sb), o, mask, em, policy=pol)
        fn(); torch.cuda.synchronize()
        res[pol] = (o, bench._prewarmed_us(fn, 20, 100.0))
    os_ = torch.zeros((g, mmax, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((qa, sa), (qb, sb), os_, mask, em, strict=True, sync=True)
    ul = (key(res["bf16_exact_ue8m0"][0]) - key(os_)).abs()
    print(f"grouped 256 x (128, 7168, 2048) mask {name}: " + "  ".join(f"{p} {v[1]:.1f} us" for p, v in res.items()) +
          f" | bx_ue8m0 vs strict max_ulp {int(ul.max())} frac>2ulp {float((ul > 2).double().mean()):.2e}" +
          (f" | full-mask frac of 8 TB/s: " + "  ".join(f"{p} {alg / v[1] / 8e6:.3f}" for p, v in res.items()) if name == "full" else ""), flush=True)
