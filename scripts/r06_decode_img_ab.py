"""Development build only (make FLAGS_dga_launch_menu_n=-DDGA_DSK_KNOBS): the decode split-K with the A tile converted once per k group
(the image build, default) against the build that converts it in every wave ($DGA_DSK_KNOB=64); graph replay, one process, alternating."""
import os, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench
import deepgemm_ascend_amd as dga

shapes = [tuple(int(x) for x in a.split("x")) for a in sys.argv[1:]] or [(64, 4096, 7168), (128, 4096, 7168), (64, 7168, 18432), (64, 7168, 16384), (64, 24576, 1536),
                                                                          (128, 2112, 7168), (64, 4096, 4096), (32, 24576, 1536), (256, 4096, 7168), (40, 16384, 7168), (128, 7168, 2048)]
g = torch.Generator(device="cuda").manual_seed(1)
for (m, n, k) in shapes:
    kb = -(-k // 128)
    a = torch.randint(0, 120, (m, k), dtype=torch.uint8, device="cuda", generator=g)
    b = torch.randint(0, 120, (n, k), dtype=torch.uint8, device="cuda", generator=g)
    sfa = torch.rand((m, kb), device="cuda", generator=g) + 0.5
    sfb = torch.rand((-(-n // 128), kb), device="cuda", generator=g) + 0.5
    t = dga.tiling(m, n, k, policy="bf16_exact")
    outs, row = {}, {}
    for knob in (0, 64, 0, 64):
        os.environ["DGA_DSK_KNOB"] = str(knob)
        out = torch.zeros((m, n), dtype=torch.bfloat16, device="cuda")
        f = lambda: dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, policy="bf16_exact", tiling_=t)
        f(); torch.cuda.synchronize()
        outs[knob] = out
        row.setdefault("image" if knob == 0 else "per_wave", []).append(round(bench._graph_us(f, 20), 2))
    same = bool(torch.equal(outs[0].view(torch.int16), outs[64].view(torch.int16)))
    print(f"{m}x{n}x{k} s{t.splitkFactor} b{t.build}", row, "same bits" if same else "DIFFERENT BITS", flush=True)
