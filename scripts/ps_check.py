"""Persistent loader-wave builds (dispatchPolicyTag 5) against the one-tile builds (policy 0 / 4): byte equality over masked,
indexed, contiguous, dense and odd-K problems, then timing on the BASELINE configs[3] stream (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
import bench
from widen_perf import timeit

dev = torch.device("cuda")
gen = torch.Generator(device=dev).manual_seed(7)


def rand_fp8(shape):
    return torch.randint(0, 120, shape, dtype=torch.uint8, device=dev, generator=gen) | (
        torch.randint(0, 2, shape, dtype=torch.uint8, device=dev, generator=gen) << 7)


def tiling_for(m, n, k, groups, tile, pol, contiguous=False):
    t = dga.tiling(m, n, k, groups=groups, contiguous=contiguous)
    t.m1, t.n1, t.wavesM, t.wavesN = tile
    t.stages, t.dispatchPolicyTag, t.kernelSerial, t.splitkFactor = 3, pol, 0, 1
    return t


TILES = [(128, 256, 2, 2), (128, 128, 2, 2), (64, 256, 1, 4), (64, 128, 1, 4), (16, 128, 1, 4)]
ok_all = True

# ---- masked grouped, several masks, every loader-wave tile
for (G, m_max, n, k) in [(40, 128, 2048, 1024), (9, 96, 520, 1040), (300, 64, 512, 384), (3, 128, 256, 128)]:
    kb, nb = -(-k // 128), -(-n // 128)
    a = rand_fp8((G, m_max, k)); b = rand_fp8((G, n, k))
    sfa = torch.rand((G, m_max, kb), device=dev, generator=gen) + 0.5
    sfb = torch.rand((G, nb, kb), device=dev, generator=gen) + 0.5
    for mask_name in ("full", "random", "sparse", "zero"):
        if mask_name == "full": mm = torch.full((G,), m_max, dtype=torch.int32, device=dev)
        elif mask_name == "random": mm = torch.randint(0, m_max + 1, (G,), dtype=torch.int32, device=dev, generator=gen)
        elif mask_name == "sparse": mm = torch.where(torch.rand((G,), device=dev, generator=gen) < 0.7, 0, 5).to(torch.int32)
        else: mm = torch.zeros((G,), dtype=torch.int32, device=dev)
        for tile in TILES:
            outs = {}
            for pol in (0, 5):
                t = tiling_for(m_max, n, k, G, tile, pol)
                o = torch.full((G, m_max, n), -1.0, dtype=torch.bfloat16, device=dev)
                for _ in range(2):
                    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o, mm, m_max, tiling_=t, sync=True)
                outs[pol] = o
            same = torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
            ok_all &= same
            if not same:
                print("MISMATCH masked", G, m_max, n, k, mask_name, tile, flush=True)
print("masked grouped: done", ok_all, flush=True)

# ---- dense (groups 1), including N / M edges and odd K
for (m, n, k) in [(4096, 2048, 1024), (300, 520, 1040), (128, 256, 128), (1000, 4096, 384), (77, 130, 200)]:
    kb, nb = -(-k // 128), -(-n // 128)
    a = rand_fp8((m, k)); b = rand_fp8((n, k))
    sfa = torch.rand((m, kb), device=dev, generator=gen) + 0.5
    sfb = torch.rand((nb, kb), device=dev, generator=gen) + 0.5
    for tile in TILES:
        outs = {}
        for pol in (0, 5):
            t = tiling_for(m, n, k, 1, tile, pol)
            o = torch.full((m, n), -1.0, dtype=torch.bfloat16, device=dev)
            for _ in range(2):
                dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), o, tiling_=t, sync=True)
            outs[pol] = o
        same = torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
        ok_all &= same
        if not same:
            print("MISMATCH dense", m, n, k, tile, flush=True)
print("dense: done", ok_all, flush=True)

# ---- contiguous layout
for (groups, n, k) in [(5, 512, 640), (8, 4096, 1024)]:
    kb, nb = -(-k // 128), -(-n // 128)
    seg = [128 * int(x) for x in torch.randint(0, 4, (groups,))]
    m = sum(seg) + 128
    idx = torch.full((m,), -1, dtype=torch.int32)
    pos = 0
    for g, s in enumerate(seg):
        idx[pos:pos + s] = g
        if s: idx[pos + s - 17:pos + s] = -1   # padding rows at the end of the segment
        pos += s
    idx = idx.to(dev)
    a = rand_fp8((m, k)); b = rand_fp8((groups, n, k))
    sfa = torch.rand((m, kb), device=dev, generator=gen) + 0.5
    sfb = torch.rand((groups, nb, kb), device=dev, generator=gen) + 0.5
    for tile in TILES:
        outs = {}
        for pol in (0, 5):
            t = tiling_for(m, n, k, groups, tile, pol, contiguous=True)
            o = torch.full((m, n), -1.0, dtype=torch.bfloat16, device=dev)
            for _ in range(2):
                dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), o, idx, tiling_=t, sync=True)
            outs[pol] = o
        same = torch.equal(outs[0].view(torch.int16), outs[5].view(torch.int16))
        ok_all &= same
        if not same:
            print("MISMATCH contiguous", groups, n, k, tile, flush=True)
print("contiguous: done", ok_all, flush=True)
print("ALL EQUAL" if ok_all else "FAILED", flush=True)
if not ok_all:
    sys.exit(1)

# ---- timing: BASELINE configs[3]
G, m_max, n, k = 256, 128, 2048, 7168
kb, nb = k // 128, n // 128
a = rand_fp8((G, m_max, k)); b = rand_fp8((G, n, k))
sfa = torch.rand((G, m_max, kb), device=dev, generator=gen) + 0.5
sfb = torch.rand((G, nb, kb), device=dev, generator=gen) + 0.5
o = torch.empty((G, m_max, n), dtype=torch.bfloat16, device=dev)
for mask_name in ("full", "random", "decode32"):
    if mask_name == "full": mm = torch.full((G,), m_max, dtype=torch.int32, device=dev)
    elif mask_name == "random": mm = torch.randint(0, m_max + 1, (G,), dtype=torch.int32, device=dev, generator=gen)
    else: mm = torch.randint(0, 33, (G,), dtype=torch.int32, device=dev, generator=gen)
    line = f"C4 {mask_name}:"
    for pol in (0, 4, 5, 4, 5):
        t = tiling_for(m_max, n, k, G, (128, 256, 2, 2), pol)
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), o, mm, m_max, tiling_=t)
        us = min(timeit(fn, iters=30, warm=30) for _ in range(3))
        line += f"  p{pol} {us:.1f} us"
    print(line, flush=True)
# dense shapes with several tiles per CU
for (m, n, k) in [(8192, 4096, 4096), (4096, 7168, 2048), (2048, 7168, 4096), (1024, 18432, 7168), (4096, 2048, 7168)]:
    a2, sfa2, b2, sfb2 = bench.make_dense_inputs(m, n, k, seed=0)
    o2 = torch.empty((m, n), dtype=torch.bfloat16, device=dev)
    line = f"dense {m}x{n}x{k}:"
    t0 = dga.select_kernel(m, n, k)
    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a2, sfa2), (b2, sfb2), o2, tiling_=t0)
    line += f"  heuristic({t0.m1}x{t0.n1} p{t0.dispatchPolicyTag}) {min(timeit(fn, iters=40, warm=60) for _ in range(3)):.1f}"
    for pol in (4, 5):
        t = tiling_for(m, n, k, 1, (128, 256, 2, 2), pol)
        fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a2, sfa2), (b2, sfb2), o2, tiling_=t)
        line += f"  128x256 p{pol} {min(timeit(fn, iters=40, warm=60) for _ in range(3)):.1f}"
    print(line, flush=True)
