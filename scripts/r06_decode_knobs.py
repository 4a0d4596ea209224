"""Development build only (make -C deepgemm_ascend_amd/csrc FLAGS_dga_launch_menu_n=-DDGA_DSK_KNOBS): where the one-launch decode split-K
spends its time.  $DGA_DSK_KNOB bits: 1 = partials neither written nor awaited, 2 = written, not awaited, 4 = nothing multiplied (the refill alone),
8 = nothing fetched, 32 = the refill in a burst behind the barrier, 64 = the build that converts A in every wave (4, 8, 32 act on that build
only).  Graph replay, one process.  Numbers of record: DESIGN.md 2.6."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd import _lib

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(64, 4096, 7168), (128, 4096, 7168)]
g = torch.Generator(device="cuda").manual_seed(1)
for (m, n, k) in shapes:
    kb = -(-k // 128)
    a = torch.randint(0, 127, (m, k), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 127, (n, k), dtype=torch.uint8, device="cuda", generator=g)
    sfa = torch.rand((m, kb), device="cuda", generator=g) + 0.5
    sfb = torch.rand((-(-n // 128), kb), device="cuda", generator=g) + 0.5
    out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
    tiles = -(-m // 64) * -(-n // 128)
    for s in (1, 3, 4, 6, 7):
        if tiles * s > 256 or kb < 4 * s:
            continue
        t = dga.tiling(m, n, k, policy="bf16_exact")
        t.m1, t.n1, t.kernelSerial, t.build, t.splitkFactor, t.stages = 64, 128, 6, _lib.BUILD_BX_DECODE, s, 0
        f = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
        row = {}
        for knob in (0, 64, 65, 66, 69, 73, 77, 96):
            os.environ["DGA_DSK_KNOB"] = str(knob)
            f(); torch.cuda.synchronize()
            row[f"k{knob}"] = round(bench._graph_us(f, 20), 2)
        os.environ["DGA_DSK_KNOB"] = "0"
        print(f"{m}x{n}x{k} S={s}", row, flush=True)
