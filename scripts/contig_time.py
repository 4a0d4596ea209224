import sys; sys.path.insert(0, sys.argv[1] if len(sys.argv) > 1 else "/root/repo")
import torch, time
import deepgemm_ascend_amd as dga
gen = torch.Generator(device="cuda").manual_seed(7)
groups, per, n, k = 8, 1024, 4096, 7168
msum = groups * per
a = torch.randint(0, 120, (msum, k), dtype=torch.uint8, device="cuda", generator=gen); b = torch.randint(0, 120, (groups, n, k), dtype=torch.uint8, device="cuda", generator=gen)
sfa = torch.rand((msum, k // 128), device="cuda") + 0.5
sfb = torch.rand((groups, n // 128, k // 128), device="cuda") + 0.5
idx = torch.arange(groups, device="cuda", dtype=torch.int32).repeat_interleave(per).contiguous()
out = torch.empty((msum, n), dtype=torch.bfloat16, device="cuda")
t = dga.tiling(msum, n, k, groups=groups, contiguous=True)
fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(20): fn()
    torch.cuda.synchronize()
e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): fn()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 50
print("tile", t.m1, t.n1, "policy", t.dispatchPolicyTag, "stages", t.stages, "us %.1f" % us, "TF %.0f" % (2.0 * msum * n * k / us / 1e6))
