"""Characterise v_mfma_f32_16x16x128_f8f6f4's internal accumulation (development aid)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import deepgemm_ascend_amd as dga
from oracle import oracle as O

enc = lambda v: O.lib().dga_oracle_f32_to_e4m3fn(float(v))
def run(arow, brow):
    k = len(arow)
    a = np.zeros((16, k), np.uint8); b = np.zeros((128, k), np.uint8)
    a[0] = [enc(v) for v in arow]; b[0] = [enc(v) for v in brow]
    sfa = np.ones((16, (k+127)//128), np.float32); sfb = np.ones((1, (k+127)//128), np.float32)
    out = torch.zeros((16, 128), dtype=torch.bfloat16, device="cuda")
    # bf16 output hides low bits: subtract the big term via a second k-block? keep simple: use f32 compare by making
    # the big product cancel: a = [X, -X, small...] so exact = sum(small)
    dga.gemm_fp8_fp8_bf16_nt((torch.from_numpy(a).cuda(), torch.from_numpy(sfa).cuda()),
                             (torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda()), out, sync=True)
    return float(out[0, 0].float().cpu())

print("cancel test: a=[X,-X,v*126], b=[X,X,w*126] -> exact 126*v*w")
for X in [448, 256, 64, 16, 1]:
    for vw in [(1,1),(0.5,0.5),(0.125,0.125),(2**-6,2**-6),(2**-6, 2**-9)]:
        v,w = vw
        r = run([X,-X]+[v]*126, [X,X]+[w]*126)
        print(f"  X={X:4} v*w={v*w:.3e} exact={126*v*w:.6e} got={r:.6e}")
print("position test: big pair at positions (p, p+1)")
for p in [0, 15, 16, 31, 32, 63, 64, 100, 126]:
    ar = [0.125]*128; br=[0.125]*128
    ar[p]=448; ar[p+1]=-448; br[p]=448; br[p+1]=448
    print(f"  p={p} exact={126*0.125*0.125:.6e} got={run(ar,br):.6e}")
print("big pair split across halves (0 and 64), (0 and 32), (0 and 16)")
for q in [16, 32, 64, 127]:
    ar = [0.125]*128; br=[0.125]*128
    ar[0]=448; ar[q]=-448; br[0]=448; br[q]=448
    print(f"  q={q} got={run(ar,br):.6e}")
print("single big + smalls, no cancellation: X*X + 127*v*w (bf16 out, so only coarse)")
for vw in [1, 4, 16]:
    r = run([448]+[vw]*127, [448]+[1]*127)
    print(f"  vw={vw} exact={448*448+127*vw} got={r}")
