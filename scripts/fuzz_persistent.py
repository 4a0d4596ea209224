"""Random masked / contiguous / dense problems: the persistent builds (dispatchPolicyTag 5, 6) against the one-tile builds,
byte for byte (development aid; run under `timeout`: a mismatched barrier count would hang)."""
import sys, random
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 150
rng = random.Random(seed)
dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(seed)
TILES = [(128, 256, 2, 2), (128, 128, 2, 2), (64, 256, 1, 4), (64, 128, 1, 4), (16, 128, 1, 4)]
bad = 0


def rf(shape):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device=dev, generator=gen)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)


for it in range(cases):
    kind = rng.choice(["masked", "masked", "dense", "dense6", "contiguous"])
    k = rng.choice([128, 256, 384, 400, 512, 640, 1000, 1024])
    kb = -(-k // 128)
    if k % 16:
        k = k // 16 * 16
        kb = -(-k // 128)
    if kind == "masked":
        g, mm, n = rng.randint(1, 400), rng.choice([16, 48, 64, 100, 128, 200]), rng.choice([128, 256, 392, 512, 1000, 2048])
        tile = rng.choice([t for t in TILES if t[0] <= max(16, mm)] or TILES[-1:])
        a, b = rf((g, mm, k)), rf((g, n, k))
        sfa = torch.rand((g, mm, kb), device=dev, generator=gen) + 0.5
        sfb = torch.rand((g, -(-n // 128), kb), device=dev, generator=gen) + 0.5
        mask = torch.randint(0, mm + 1, (g,), dtype=torch.int32, device=dev, generator=gen)
        mask[torch.rand((g,), device=dev, generator=gen) < rng.choice([0.0, 0.3, 0.9])] = 0
        outs = []
        for pol in (0, 5):
            t = dga.tiling(mm, n, k, groups=g, expected_m=mm)
            t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = *tile, 3, pol, 0, 1
            o = torch.full((g, mm, n), -3.0, dtype=torch.bfloat16, device=dev)
            dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o, mask, mm, tiling_=t, sync=True)
            outs.append(o)
        desc = f"masked g{g} m{mm} n{n} k{k} tile{tile}"
    elif kind == "contiguous":
        groups, n = rng.randint(1, 12), rng.choice([128, 512, 1024, 2100])
        seg = [128 * rng.randint(0, 3) for _ in range(groups)]
        m = sum(seg) + 128 * rng.randint(0, 2)
        if m == 0:
            continue
        idx = torch.full((m,), -1, dtype=torch.int32)
        pos = 0
        for gi, sg in enumerate(seg):
            idx[pos:pos + sg] = gi
            if sg:
                idx[pos + sg - rng.randint(0, 40):pos + sg] = -1
            pos += sg
        idx = idx.to(dev)
        tile = rng.choice(TILES)
        a, b = rf((m, k)), rf((groups, n, k))
        sfa = torch.rand((m, kb), device=dev, generator=gen) + 0.5
        sfb = torch.rand((groups, -(-n // 128), kb), device=dev, generator=gen) + 0.5
        outs = []
        for pol in (0, 5):
            t = dga.tiling(m, n, k, groups=groups, contiguous=True)
            t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = *tile, 3, pol, 0, 1
            o = torch.full((m, n), -3.0, dtype=torch.bfloat16, device=dev)
            dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), o, idx, tiling_=t, sync=True)
            outs.append(o)
        desc = f"contiguous groups{groups} m{m} n{n} k{k} tile{tile}"
    else:
        if kind == "dense6":
            m, n = 256 * rng.randint(1, 24), 256 * rng.randint(1, 24)
            k = 128 * rng.randint(2, 9); kb = k // 128
            tile, pols, st = (256, 256, 4, 2), (2, 6), 2
        else:
            m, n = rng.randint(1, 5000), rng.randint(1, 5000)
            tile, pols, st = rng.choice(TILES), (0, 5), 3
        a, b = rf((m, k)), rf((n, k))
        sfa = torch.rand((m, kb), device=dev, generator=gen) + 0.5
        sfb = torch.rand((-(-n // 128), kb), device=dev, generator=gen) + 0.5
        outs = []
        for pol in pols:
            t = dga.tiling(m, n, k)
            t.m1, t.n1, t.wavesM, t.wavesN, t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = *tile, st, pol, 0, 1
            o = torch.full((m, n), -3.0, dtype=torch.bfloat16, device=dev)
            dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t, sync=True)
            outs.append(o)
        desc = f"{kind} m{m} n{n} k{k} tile{tile}"
    same = torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    if not same:
        bad += 1
        print("MISMATCH", desc, flush=True)
    if it % 25 == 0:
        print(f"case {it}: {desc} ok={same}", flush=True)
print(f"{cases} cases, {bad} mismatches", flush=True)
sys.exit(1 if bad else 0)
