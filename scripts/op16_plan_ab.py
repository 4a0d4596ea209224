"""The aclnn operator's 16-bit path: the plan's tile against the alternatives on mid-size shapes ($DGA_B16_PLAN forces a plan;
one subprocess per plan, device time by graph replay)."""
import os
os.environ.setdefault("DGA_B16_DEV", "1")   # the 16-bit operators read their development switches per call only when told so (dga_b16.hip)
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
CHILD = r'''
import sys; sys.path.insert(0, %r)
import torch
import deepgemm_ascend_amd as dga
from deepgemm_ascend_amd.harness import sweep
for s in sys.argv[1:]:
    m, n, k = (int(x) for x in s.split(","))
    x = torch.randn(m, k, device="cuda", dtype=torch.bfloat16); y = torch.randn(n, k, device="cuda", dtype=torch.bfloat16)
    o = torch.empty(m, n, device="cuda", dtype=torch.bfloat16)
    fn = lambda: dga.catlass_dynamic_matmul(x, y.t(), o)
    fn(); torch.cuda.synchronize()
    ref = x.float() @ y.float().t()
    err = float((o.float() - ref).abs().max() / ref.abs().max())
    print(s, "%%.1f" %% sweep.graph_us(fn, 10, 3, 30.0), "%%.1e" %% err, flush=True)
''' % str(ROOT)
shapes = sys.argv[1].split() if len(sys.argv) > 1 else ["1024,4096,7168", "1024,18432,7168", "2048,4096,7168", "4096,4096,4096", "512,7168,4096", "3072,4096,4096", "1536,6144,4096", "768,8192,8192", "4608,4096,7168"]
res = {}
for plan in ((None,) + tuple(sys.argv[2].split())) if len(sys.argv) > 2 else (None, "256,256,1", "128,256,1", "128,128,1", "128,256,2", "128,128,2", "256,256,2"):
    env = dict(os.environ)
    if plan:
        env["DGA_B16_PLAN"] = plan
    r = subprocess.run([sys.executable, "-c", CHILD] + shapes, env=env, capture_output=True, text=True, timeout=600)
    for line in r.stdout.splitlines():
        parts = line.split()
        if len(parts) == 3:
            res.setdefault(parts[0], {})[plan or "auto"] = (float(parts[1]), parts[2])
for s in shapes:
    row = res.get(s, {})
    print(s.ljust(18), "  ".join(f"{p}: {v[0]:7.1f}" for p, v in row.items()), flush=True)
