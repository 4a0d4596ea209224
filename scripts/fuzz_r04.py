"""Randomised bit-identity campaign for round 4's kernels (each against the path it would replace), through the operator:
  image     bf16-exact image builds (8 and 4 waves, A-image) == the in-register build   dense, any M N K % 16 == 0
  strided   row-strided / aligned-row operands, every policy == the contiguous call     any M N K, strides, promises
  wsk       one-launch workgroup split-K (kernelSerial 6) == two-launch split-K 8       M <= 64
  unaligned odd K read in place (kernelSerial 2) == padding pass + the same tile        K % 16 != 0
Usage: python scripts/fuzz_r04.py [cases per kernel = 150] [seed = 0]; exit code 1 on the first mismatch (the case is printed)."""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import deepgemm_ascend_amd as dga  # noqa: E402


def data(m, n, k, gen, nan=False):
    a = torch.randint(0, 256, (m, k), dtype=torch.uint8, device="cuda", generator=gen)
    b = torch.randint(0, 256, (n, k), dtype=torch.uint8, device="cuda", generator=gen)
    if not nan:
        a[(a & 0x7F) == 0x7F] = 0x3C; b[(b & 0x7F) == 0x7F] = 0x3C
    kb = (k + 127) // 128
    sfa = torch.exp2(torch.rand((m, kb), device="cuda", generator=gen) * 8 - 4)
    sfb = torch.exp2(torch.rand(((n + 127) // 128, kb), device="cuda", generator=gen) * 8 - 4)
    return a, sfa, b, sfb


def run(a, sfa, b, sfb, t, **kw):
    out = torch.full((a.shape[0], b.shape[0]), float("nan"), dtype=torch.bfloat16, device="cuda")
    dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, sync=True, tiling_=t, **kw)
    return out.view(torch.int16)


def same(x, y):   # bit identity; NaNs compare by position (payloads of the two paths may differ)
    xn, yn = (x & 0x7FFF) > 0x7F80, (y & 0x7FFF) > 0x7F80
    return bool(torch.equal(xn, yn) and torch.equal(x[~xn], y[~yn]))


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    gen = torch.Generator(device="cuda").manual_seed(seed)
    done = {"image": 0, "wsk": 0, "unaligned": 0}
    for i in range(cases):
        # ---- image builds
        m, n = int(rng.integers(1, 700)), int(rng.integers(1, 1200))
        k = int(rng.integers(1, 80)) * 16
        a, sfa, b, sfb = data(m, n, k, gen, nan=(i % 17 == 0))
        sk = int(rng.choice([1, 1, 2, 3]))
        raster = int(rng.choice([1, 2, 4, 8]))
        outs = []
        for image in (0, 8, 4, 1):
            t = dga.tiling(m, n, k)
            t.m1, t.n1, t.splitkFactor, t.kernelSerial, t.dispatchPolicyTag = 128, 256, sk, (4 if sk > 1 else 0), 7
            t.stages, t.swizzleOffset = {0: 3, 1: 4, 8: 5, 4: 6}[image], raster
            t.wavesM, t.wavesN = 2, 4
            outs.append(run(a, sfa, b, sfb, t, policy="bf16_exact"))
        if not (same(outs[0], outs[1]) and same(outs[0], outs[2]) and same(outs[0], outs[3])):
            print(f"MISMATCH image: m={m} n={n} k={k} splitk={sk} raster={raster} case {i} seed {seed}")
            return 1
        done["image"] += 1
        # ---- workgroup split-K
        m = int(rng.integers(1, 65)); n = int(rng.integers(1, 3000)); k = int(rng.integers(1, 600)) * 16
        if i % 3 == 0:   # more than one n-tile per workgroup: the multi-pass LDS-DMA builds (M <= 32)
            n = int(rng.integers(4097, 15000)); k = int(rng.integers(1, 200)) * 16
        a, sfa, b, sfb = data(m, n, k, gen)
        t6 = dga.tiling(m, n, k); t6.kernelSerial, t6.splitkFactor = 6, 1
        t4 = dga.tiling(m, n, k); t4.kernelSerial, t4.splitkFactor = 4, 8
        t4.m1, t4.n1, t4.stages, t4.wavesM, t4.wavesN, t4.dispatchPolicyTag = 64, 128, 3, 1, 4, 0
        if not same(run(a, sfa, b, sfb, t6), run(a, sfa, b, sfb, t4)):
            print(f"MISMATCH wsk: m={m} n={n} k={k} case {i} seed {seed}")
            return 1
        if m <= 32 and i % 2:   # the same rings with the bf16-exact policy's arithmetic == that policy's two-launch split-K 8
            if not same(run(a, sfa, b, sfb, t6, policy="bf16_exact"), run(a, sfa, b, sfb, t4, policy="bf16_exact")):
                print(f"MISMATCH wsk bf16-exact: m={m} n={n} k={k} case {i} seed {seed}")
                return 1
        done["wsk"] += 1
        # ---- odd K in place
        m, n = int(rng.integers(1, 600)), int(rng.integers(1, 900))
        k = int(rng.integers(1, 3000))
        if k % 16 == 0:
            k += int(rng.integers(1, 16))
        a, sfa, b, sfb = data(m, n, k, gen, nan=(i % 19 == 0))
        res = []
        for ks in (2, 0):
            t = dga.tiling(m, n, k)
            t.m1, t.n1, t.stages, t.wavesM, t.wavesN, t.splitkFactor, t.dispatchPolicyTag, t.kernelSerial = 128, 256, 3, 2, 2, 1, 4, ks
            res.append(run(a, sfa, b, sfb, t))
        if not same(res[0], res[1]):
            print(f"MISMATCH unaligned: m={m} n={n} k={k} case {i} seed {seed}")
            return 1
        done["unaligned"] += 1
        # ---- row-strided operands (this case's odd K, and a K % 16 == 0 one every other round), auto tiling, a policy per round
        if i % 2:
            k = (k + 15) // 16 * 16
            a, sfa, b, sfb = data(m, n, k, gen)
        kw = [{}, {"policy": "bf16_exact"}, {"strict": True}][i % 3]
        ref = run(a, sfa, b, sfb, None, **kw)
        lds = [((k + 15) // 16 + int(rng.integers(0, 9))) * 16 for _ in range(2)]
        views = []
        for x, ld in ((a, lds[0]), (b, lds[1])):
            buf = torch.randint(0, 256, (x.shape[0], ld), dtype=torch.uint8, device="cuda", generator=gen)
            buf[:, :k] = x
            buf[:, k:(k + 15) // 16 * 16] = 0
            views.append(buf[:, :k])
        zp = [(True, True), (False, False), (True, False), (False, True)][int(rng.integers(0, 4))]
        ops = [views[j] if (zp[j] or rng.integers(0, 2)) else (a, b)[j] for j in range(2)]
        if not same(run(ops[0], sfa, ops[1], sfb, None, zero_padded=zp, **kw), ref):
            print(f"MISMATCH strided: m={m} n={n} k={k} lds={lds} zero_padded={zp} policy={kw} case {i} seed {seed}")
            return 1
        done["strided"] = done.get("strided", 0) + 1
        if i % 25 == 24:
            print(f"{i + 1} rounds: {done}", flush=True)
    print(f"fuzz ok: {done} (seed {seed})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
