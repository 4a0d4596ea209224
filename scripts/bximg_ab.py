"""A/B of the bf16-exact policy's two builds of the 128 x 256 tile -- the image build (one wave per SIMD, operands converted
once per workgroup into a bf16 LDS image) against the in-register build (two waves per SIMD, every wave converts what it
multiplies) -- on BASELINE configs[1] and configs[2], in one process at sustained clocks, interleaved.
Usage: python scripts/bximg_ab.py [--rasters]"""
import json
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
import deepgemm_ascend_amd as dga  # noqa: E402
from scripts.policy_perf import time_us  # noqa: E402


def tiling(m, n, k, image, raster=4):
    t = dga.tiling(m, n, k)
    t.m1, t.n1, t.splitkFactor, t.kernelSerial = 128, 256, 1, 0
    t.dispatchPolicyTag, t.stages, t.swizzleOffset = 7, {0: 3, 1: 4, 8: 5, 4: 6}[image], raster
    t.wavesM, t.wavesN = 2, 4
    return t


def main():
    res = {}
    for name in ("dense_4096", "dsv3_prefill"):
        m, n, k = bench.WORKLOADS[name]
        a, sfa, b, sfb = bench.make_dense_inputs(m, n, k, seed=0)
        out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
        row = {}
        for rep in range(2):
            for image in (0, 8, 4, 1):
                for raster in ((1, 2, 4, 8, 16) if "--rasters" in sys.argv else (4,)):
                    t = tiling(m, n, k, image, raster)
                    fn = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
                    us = time_us(fn, 200, 400)
                    key = f"{('a_image' if image == 1 else 'image%d' % image) if image else 'in_register'}_r{raster}"
                    row.setdefault(key, []).append(round(us, 2))
                    print(name, key, f"{us:.2f} us  {2.0 * m * n * k / us / 1e6:.0f} TFLOP/s", flush=True)
        res[name] = row
    print(json.dumps(res))


if __name__ == "__main__":
    main()
