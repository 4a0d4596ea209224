"""Where the end-to-end time of the expert-sharded grouped GEMM goes at world 1 (development aid)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from deepgemm_ascend_amd import parallel

dev = torch.device("cuda", 0)
for G, per in ((256, 128), (32, 128)):
    eng = parallel.ExpertShardedGroupedGemm(0, 1, G, 128, 2048, 7168, dev, None)
    g = torch.Generator(device=dev).manual_seed(1)
    eng.set_weights(parallel._rand_fp8((G, 2048, 7168), g, dev), torch.rand((G, 16, 56), device=dev) + 0.5)
    ids = torch.arange(G, device=dev).repeat_interleave(per)
    ids = ids[torch.randperm(ids.numel(), device=dev)]
    T = ids.numel()
    q = parallel._rand_fp8((T, 7168), g, dev); sf = torch.rand((T, 56), device=dev) + 0.5
    def timed(fn, n=20):
        for _ in range(3): r = fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): r = fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, r
    td, st = timed(lambda: eng.dispatch(q, sf, ids))
    tg, _ = timed(lambda: eng.run_local())
    tc, _ = timed(lambda: eng.combine(st))
    tf, _ = timed(lambda: eng.forward(q, sf, ids))
    print(f"G={G} T={T}: dispatch {td:.3f} ms  gemm {tg:.3f} ms  combine {tc:.3f} ms  forward {tf:.3f} ms", flush=True)
