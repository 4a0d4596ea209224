import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import deepgemm_ascend_amd as dga
from oracle import oracle as O

def gpu(a, sfa, b, sfb):
    out = torch.zeros((a.shape[0], b.shape[0]), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
                             (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()), out, sync=True)
    return out.view(torch.int16).cpu().numpy().view(np.uint16)

worst_all = 0
for (m, n, k, seed, mode) in [(512, 512, 128, 1, "normal"), (512, 512, 512, 2, "normal"), (256, 512, 4096, 3, "normal"),
                              (256, 512, 7168, 4, "normal"), (512, 512, 128, 5, "unit"), (512, 512, 1024, 6, "bits"),
                              (512, 512, 128, 7, "bits"), (1024, 1024, 256, 8, "normal")]:
    if mode == "bits":
        rng = np.random.default_rng(seed)
        a = O.random_fp8_bytes((m, k), seed=seed); b = O.random_fp8_bytes((n, k), seed=seed + 100)
        sfa = np.exp2(rng.uniform(-8, 4, size=(m, (k+127)//128))).astype(np.float32)
        sfb = np.exp2(rng.uniform(-8, 4, size=((n+127)//128, (k+127)//128))).astype(np.float32)
    else:
        a, sfa, b, sfb = O.make_inputs(m, n, k, seed=seed, unit_scales=(mode == "unit"))
    got = gpu(a, sfa, b, sfb)
    want = O.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=16)
    ok, worst = O.parity_excess(got, want, a, sfa, b, sfb)
    d = O.bf16_ulp_diff(got, want)
    worst_all = max(worst_all, worst)
    print(f"{m}x{n}x{k} {mode}: plain max ulp {d.max()}, >2ulp: {(d>2).sum()}/{d.size}, worst excess/S = {worst:.3e} = 2^{np.log2(max(worst,1e-30)):.2f}", flush=True)
print("overall worst", worst_all, np.log2(worst_all))
