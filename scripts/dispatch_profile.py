"""Host/device breakdown of ExpertShardedGroupedGemm.dispatch at the per-rank size of the 8-GPU config (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from torch.profiler import profile, ProfilerActivity
from deepgemm_ascend_amd import parallel
dev = torch.device("cuda", 0)
G, per = 32, 128
eng = parallel.ExpertShardedGroupedGemm(0, 1, G, 128, 2048, 7168, dev, None)
g = torch.Generator(device=dev).manual_seed(1)
eng.set_weights(parallel._rand_fp8((G, 2048, 7168), g, dev), torch.rand((G, 16, 56), device=dev) + 0.5)
ids = torch.arange(G, device=dev).repeat_interleave(per); ids = ids[torch.randperm(ids.numel(), device=dev)]
q = parallel._rand_fp8((ids.numel(), 7168), g, dev); sf = torch.rand((ids.numel(), 56), device=dev) + 0.5
for _ in range(5): eng.dispatch(q, sf, ids)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10): eng.dispatch(q, sf, ids)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
