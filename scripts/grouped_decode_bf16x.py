"""The bf16-exact policy on the masked grouped layout with nearly empty experts (decode MoE): BASELINE configs[3]'s shape, rows present
0..16 / 0..32 / 0..64 / 0..128 per expert, by tile height -- the policy's own pick with and without the expected_m hint, and forced tiles.
Usage: python scripts/grouped_decode_bf16x.py"""
import json, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import deepgemm_ascend_amd as dga
from scripts.policy_perf import time_us
G, MMAX, N, K = 256, 128, 2048, 7168
g = torch.Generator(device="cuda").manual_seed(0)
a = torch.randint(0, 120, (G, MMAX, K), dtype=torch.uint8, device="cuda", generator=g)
b = torch.randint(0, 120, (G, N, K), dtype=torch.uint8, device="cuda", generator=g)
sfa = torch.rand((G, MMAX, K // 128), device="cuda") + 0.5
sfb = torch.rand((G, N // 128, K // 128), device="cuda") + 0.5
out = torch.zeros((G, MMAX, N), dtype=torch.bfloat16, device="cuda")
for mask, masked in (("0..16", torch.randint(0, 17, (G,), dtype=torch.int32, device="cuda", generator=g)),
                     ("0..32", torch.randint(0, 33, (G,), dtype=torch.int32, device="cuda", generator=g)),
                     ("0..64", torch.randint(0, 65, (G,), dtype=torch.int32, device="cuda", generator=g)),
                     ("random", torch.randint(0, 129, (G,), dtype=torch.int32, device="cuda", generator=g))):
    rows = int(masked.sum())
    byt = G * N * K + rows * (K + 4 * (K // 128) + 2 * N)
    hint = {"0..16": 16, "0..32": 32, "0..64": 64}.get(mask, MMAX)
    for pol, tile in (("fast", None), ("bf16_exact", None), ("bf16_exact", "hint"), ("bf16_exact", (64, 256)), ("bf16_exact", (32, 128)), ("bf16_exact", (128, 128))):
        kw = {} if pol == "fast" else {"policy": pol}
        t = dga.tiling(MMAX, N, K, groups=G, expected_m=(hint if tile == "hint" else MMAX), **({"policy": pol} if pol != "fast" else {}))
        if tile and tile != "hint": t.m1, t.n1 = tile
        fn = lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked, MMAX, tiling_=t, **kw)
        us = time_us(fn, 20, 100)
        print(mask, pol, (f"expected_m={hint} -> {t.m1}x{t.n1}" if tile == "hint" else tile) or (t.m1, t.n1), round(us, 1), "us", round(byt / us / 8e6, 3), "of 8 TB/s", flush=True)
