"""Launch run_mmad_rtc bf16 a few times (for rocprofv3 runs)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import deepgemm_ascend_amd as dga
m, n, k = (int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (4096, 4096, 4096)))
x = torch.randn((1, m, k), device="cuda").to(torch.bfloat16); y = torch.randn((1, k, n), device="cuda").to(torch.bfloat16)
z = torch.empty((1, m, n), dtype=torch.float32, device="cuda")
for _ in range(8):
    dga.run_mmad_rtc(x, y, z)
print("done")
