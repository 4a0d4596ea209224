"""The DESIGN.md results table measured in one process at sustained clocks (300 ms pre-warm, then every workload:
20 warmup launches, best of 3 timed batches).  Development aid; bench.py is the judged entry."""
import sys, time, json
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from deepgemm_ascend_amd import parallel

def timeit(fn, iters, warm=20, reps=3):
    for _ in range(warm): fn()
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

rows = []
def report(name, us, flops=None, byts=None, cold_us=None, sets=None):
    r = {"workload": name, "us": round(us, 1)}
    if flops: r["TFLOP/s"] = round(flops / us / 1e6, 1)
    if byts: r["GB/s"] = round(byts / us / 1e3, 1)
    if cold_us is not None:
        # SURVEY.md 8(d) cold-cache protocol: consecutive launches rotate over `sets` operand sets that together exceed
        # the 256 MiB Infinity Cache, so every launch streams its operands from HBM; "us" / "GB/s" above re-launch on ONE
        # set (operands resident in the Infinity Cache when they fit)
        r["cold_us"] = round(cold_us, 1); r["operand_sets"] = sets
        if byts: r["cold_GB/s"] = round(byts / cold_us / 1e3, 1)
        if flops: r["cold_TFLOP/s"] = round(flops / cold_us / 1e6, 1)
    rows.append(r); print(json.dumps(r), flush=True)

def timeit_rotating(fns, iters, warm=20, reps=3):
    """Like timeit, but launch i runs fns[i % len(fns)] (each bound to its own operand set)."""
    n = len(fns)
    for i in range(warm): fns[i % n]()
    best = 1e30
    for _ in range(reps):
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(iters): fns[i % n]()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best

# pre-warm
a, sfa, b, sfb = bench.make_dense_inputs(4096, 4096, 4096, seed=0)
out = torch.empty((4096, 4096), dtype=torch.bfloat16, device="cuda")
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.3:
    for _ in range(50): dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)
    torch.cuda.synchronize()
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 2048, 7168), (1024, 18432, 7168), (1024, 4096, 7168), (128, 4096, 7168), (64, 7168, 18432), (8, 18432, 7168)]:
    a, sfa, b, sfb = bench.make_dense_inputs(max(m, 128) if m % 128 else m, n, k, seed=0)
    a, sfa = a[:m].contiguous(), sfa[:m].contiguous()
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)
    us = timeit(lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t), iters=100 if m * n * k < 2 ** 37 else 30)
    cold_us = sets = None
    opbytes = m * k + n * k + 2 * m * n
    if opbytes < 256 * 2 ** 20:   # the operands of one launch fit the Infinity Cache: also measure with rotating sets
        sets = max(3, -(-320 * 2 ** 20 // opbytes))
        copies = [(a.clone(), sfa.clone(), b.clone(), sfb.clone(), torch.empty_like(out)) for _ in range(sets)]
        fns = [(lambda c=c: dga.gemm_fp8_fp8_bf16_nt((c[0], c[1]), (c[2], c[3]), c[4], tiling_=t)) for c in copies]
        cold_us = timeit_rotating(fns, iters=max(100, 4 * sets))
        del copies, fns
    report(f"fp8 dense {m}x{n}x{k} tile {t.m1}x{t.n1} serial {t.kernelSerial} splitk {t.splitkFactor}", us, 2.0 * m * n * k, opbytes,
           cold_us=cold_us, sets=sets)
# grouped
g = torch.Generator(device="cuda").manual_seed(0)
G, MM, N, K = 256, 128, 2048, 7168
ga = parallel._rand_fp8((G, MM, K), g, "cuda"); gb = parallel._rand_fp8((G, N, K), g, "cuda")
gsfa = torch.rand((G, MM, K // 128), device="cuda") + 0.5; gsfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
gout = torch.empty((G, MM, N), dtype=torch.bfloat16, device="cuda")
for name, mask in (("full", torch.full((G,), MM, dtype=torch.int32, device="cuda")),
                   ("random", torch.randint(0, MM + 1, (G,), device="cuda", generator=g).int())):
    us = timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((ga, gsfa), (gb, gsfb), gout, mask, MM), iters=20, warm=5)
    rows_ = int(mask.sum()); act = int((mask > 0).sum())
    report(f"grouped masked 256x(128,7168,2048) mask {name} ({rows_} rows)", us, 2.0 * N * K * rows_, act * N * K + rows_ * (K + 4 * 56 + 2 * N))
del ga, gb, gout
# contiguous
groups, per, n, k = 8, 1024, 4096, 7168
ca = parallel._rand_fp8((groups * per, k), g, "cuda"); cb = parallel._rand_fp8((groups, n, k), g, "cuda")
csfa = torch.rand((groups * per, k // 128), device="cuda") + 0.5; csfb = torch.rand((groups, n // 128, k // 128), device="cuda") + 0.5
idx = torch.arange(groups, device="cuda", dtype=torch.int32).repeat_interleave(per).contiguous()
cout = torch.empty((groups * per, n), dtype=torch.bfloat16, device="cuda")
us = timeit(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((ca, csfa), (cb, csfb), cout, idx), iters=20, warm=5)
report("contiguous 8 x 1024 rows N=4096 K=7168", us, 2.0 * groups * per * n * k)
del ca, cb, cout
# quantisers
for dtype in (torch.bfloat16, torch.float32):
    x = torch.randn((32768, 7168), device="cuda").to(dtype)
    us = timeit(lambda: dga.per_token_cast_to_fp8(x), iters=20, warm=5)
    report(f"per_token_cast {dtype} [32768,7168]", us, byts=32768 * 7168 * (x.element_size() + 1))
    del x
# 16-bit paths
for (m, n, k) in [(4096, 4096, 4096), (8192, 8192, 8192)]:
    xa = torch.randn((m, k), device="cuda").bfloat16(); xb = torch.randn((n, k), device="cuda").bfloat16()
    o16 = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    us = timeit(lambda: dga.catlass_dynamic_matmul(xa, xb.t(), o16), iters=30, warm=10)
    report(f"catlass_dynamic_matmul bf16 NT {m}x{n}x{k}", us, 2.0 * m * n * k)
    y = xb.t().contiguous()[None]; z = torch.empty((1, m, n), dtype=torch.float32, device="cuda")
    us = timeit(lambda: dga.run_mmad_rtc(xa[None], y, z), iters=20, warm=5)
    report(f"run_mmad_rtc bf16 {m}x{n}x{k} (per-call sync)", us, 2.0 * m * n * k)
    del xa, xb, o16, y, z
Path(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/steady_table.json").write_text(json.dumps(rows, indent=1) + "\n")
