import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import deepgemm_ascend_amd as dga
from oracle import oracle as O

def run(m, n, k, seed, unit=False):
    a, sfa, b, sfb = O.make_inputs(m, n, k, seed=seed, unit_scales=unit)
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
                             (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()), out, sync=True)
    got = out.view(torch.int16).cpu().numpy().view(np.uint16)
    want, wf32 = O.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads=8, want_f32=True)
    ex = O.gemm_fp8_fp8_f64_nt(a, sfa, b, sfb)
    tab = O.e4m3fn_table().astype(np.float64)
    kb = (k + 127) // 128
    # sum of |terms| with scales
    absum = np.zeros((m, n))
    for i in range(kb):
        sl = slice(i * 128, min(k, i * 128 + 128))
        s = np.abs(sfa[:, i:i+1].astype(np.float64) * sfb[np.arange(n) // 128, i][None, :])
        absum += s * (np.abs(tab[a[:, sl]]) @ np.abs(tab[b[:, sl]]).T)
    gf = O.bf16_bits_to_f32(got).astype(np.float64)
    d = O.bf16_ulp_diff(got, want)
    err_gpu = np.abs(gf - ex)            # includes bf16 rounding
    err_or = np.abs(wf32.astype(np.float64) - ex)
    print(f"{m}x{n}x{k} unit={unit}: max ulp {d.max()}  count>2: {(d>2).sum()} of {d.size}")
    idx = np.argsort(d.ravel())[::-1][:5]
    for i in idx:
        r, c = np.unravel_index(i, d.shape)
        print(f"   ({r},{c}) ulp={d[r,c]} want={wf32[r,c]:.6e} got={gf[r,c]:.6e} exact={ex[r,c]:.6e} abssum={absum[r,c]:.4e} "
              f"gpu_err/abssum={abs(gf[r,c]-ex[r,c])/absum[r,c]:.3e} oracle_err/abssum={err_or[r,c]/absum[r,c]:.3e}")
    # error of GPU vs exact beyond bf16 half-ulp, normalised by abssum
    half_ulp = np.abs(ex) * 2.0 ** -9
    excess = np.maximum(err_gpu - half_ulp, 0) / absum
    print(f"   max excess err/abssum gpu: {excess.max():.3e}   oracle f32 err/abssum max: {(err_or/absum).max():.3e}")

for (m, n, k) in [(128, 128, 128), (256, 256, 512), (256, 512, 4096), (128, 256, 7168)]:
    run(m, n, k, 1)
run(128, 128, 128, 0, unit=True)
