"""Board power and shader clock (rocm-smi, read-only) while one kernel runs back to back for a few seconds: the dense 4096^3 fast path,
the same problem on the bf16-exact and strict policies, the grouped stream, and an idle baseline.  Direct evidence for "the loop runs at
constant power" (DESIGN.md section 5)."""
import json, re, subprocess, sys, threading, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga


def smi():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True, text=True, timeout=20).stdout
    try:
        d = json.loads(out)
        card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if any(s in k.lower() for s in ("power", "sclk", "junction", "edge"))}
    except Exception:
        return {"raw": out[:300]}


def run_for(fn, seconds):
    stop = [False]
    samples = []

    def sampler():
        time.sleep(seconds * 0.4)
        while not stop[0]:
            samples.append(smi())
            time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(200):
            fn()
        torch.cuda.synchronize(); n += 200
    stop[0] = True; th.join()
    return n, samples


print("idle:", smi(), flush=True)
m = n = k = 4096
a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
for name, kw, secs in (("fast 4096^3", {}, 4.0), ("bf16_exact 4096^3", {"policy": "bf16_exact"}, 4.0), ("strict 4096^3", {"strict": True}, 5.0)):
    fn = lambda kw=kw: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, **kw)
    t0 = time.perf_counter()
    cnt, samples = run_for(fn, secs)
    us = (time.perf_counter() - t0) / cnt * 1e6
    print(f"{name}: {us:.1f} us per call;", samples[-3:], flush=True)
time.sleep(1.0)
print("idle again:", smi(), flush=True)
# the 16-bit operator path and the grouped weight stream
x = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16); y = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
o = torch.empty(4096, 4096, device="cuda", dtype=torch.bfloat16)
t0 = time.perf_counter()
cnt, samples = run_for(lambda: dga.catlass_dynamic_matmul(x, y.t(), o), 4.0)
print(f"bf16 operator 4096^3: {(time.perf_counter() - t0) / cnt * 1e6:.1f} us per call;", samples[-3:], flush=True)
del x, y, o
from deepgemm_ascend_amd import parallel
gen = torch.Generator(device="cuda").manual_seed(1)
G, MM, N, K = 256, 128, 2048, 7168
gb, gsb = parallel._quantised_weights(G, N, K, gen, torch.device("cuda"))
gq, gs = parallel._quantised_tokens(G * MM, K, gen, torch.device("cuda"))
ga, gsa = gq.view(torch.uint8).view(G, MM, K), gs.view(G, MM, K // 128)
go = torch.empty((G, MM, N), dtype=torch.bfloat16, device="cuda")
mm = torch.full((G,), MM, dtype=torch.int32, device="cuda")
stop_fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((ga, gsa), (gb, gsb), go, mm, MM)


def run_few(fn, seconds):
    stop = [False]; samples = []
    def sampler():
        time.sleep(seconds * 0.4)
        while not stop[0]:
            samples.append(smi()); time.sleep(0.25)
    th = threading.Thread(target=sampler); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(20):
            fn()
        torch.cuda.synchronize(); n += 20
    stop[0] = True; th.join()
    return n, samples
t0 = time.perf_counter()
cnt, samples = run_few(stop_fn, 4.0)
print(f"grouped 256 x (128, 7168, 2048): {(time.perf_counter() - t0) / cnt * 1e6:.1f} us per call;", samples[-3:], flush=True)
