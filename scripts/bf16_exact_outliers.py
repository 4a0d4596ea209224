"""The outputs of BASELINE configs[1] on which the bf16-exact policy is more than 2 bf16 ULP from the strict kernel: how far each\nvalue has cancelled against S = the sum of the magnitudes of its scaled products (development aid behind README "Numerics")."""
import sys; sys.path.insert(0, '/root/repo')
import torch, bench, math
import deepgemm_ascend_amd as dga
m,n,k = 4096,4096,4096
a,sfa,b,sfb = bench.make_dense_inputs(m,n,k,seed=0)
bx = torch.empty((m,n),dtype=torch.bfloat16,device='cuda'); ex=torch.empty_like(bx); s=torch.empty_like(bx)
dga.gemm_fp8_fp8_bf16_nt((a,sfa),(b,sfb),bx,policy='bf16_exact')
dga.gemm_fp8_fp8_bf16_nt((a,sfa),(b,sfb),ex,strict=True)
dga.gemm_fp8_fp8_bf16_nt((a&0x7F,sfa.abs()),(b&0x7F,sfb.abs()),s,strict=True,sync=True)
def key(t):
    v=t.view(torch.int16).to(torch.int32); mag=v&0x7FFF; return torch.where(v<0,-mag,mag)
u=(key(bx)-key(ex)).abs()
idx=torch.nonzero(u>2)
print("beyond 2ulp:", idx.shape[0])
r=[]
for i,j in idx.tolist():
    r.append((int(u[i,j]), float(ex[i,j]), float(bx[i,j]), float(s[i,j]), abs(float(ex[i,j]))/float(s[i,j])))
r.sort(reverse=True)
for x in r[:8]: print(x, "log2(|v|/S)=%.1f"%math.log2(max(x[4],1e-300)))
print("largest |v|/S among them: 2^%.1f" % math.log2(max(x[4] for x in r)))
