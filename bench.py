#!/usr/bin/env python3
"""bench.py -- the judged benchmark entry (contract in the task statement, section 4).

Default workload = BASELINE.json configs[1]: 4096 x 4096 x 4096 fp8 block-scaled NT GEMM, bf16 out,
per-1x128 A scales / per-128x128 B scales, 1 x MI355X.  One "step" = one pass of the hot path over one
batch = one GEMM launch through the C ABI with inputs already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: the dense GEMM does not shard ("replicas only", DESIGN.md section e): every rank runs the same
problem, `value` = N x per-replica work / max-over-ranks time (weak scaling).  The path that does shard --
the grouped masked-M GEMM with experts partitioned over ranks and an RCCL all-to-all each way -- is reported
in the extra "grouped" object of the same JSON line (tok/s with and without the exchange).

Extra objects: "roofline" (dominant kernel vs the dense fp8 MFMA peak), "cpu_baseline" (the CPU oracle timed
on this box's host cores on a bounded row sample; rank 0, N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

PEAK_FP8_TFLOPS = 5000.0   # MI355X dense fp8 MFMA peak (/opt/skills/guides/MI355X_MICROARCH.md, chip table)
PEAK_HBM_GBPS = 8000.0

WORKLOADS = {
    # name: (m, n, k)
    "dense_4096": (4096, 4096, 4096),         # BASELINE.json configs[1]
    "dsv3_prefill": (4096, 2048, 7168),       # configs[2]  (M=4096, K=7168, N=2048)
}


def _rand_fp8(shape, gen):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda", generator=gen)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)  # no NaN encodings


def make_dense_inputs(m, n, k, seed):
    """Synthetic data of the SURVEY.md 8(d) shape: fp32 ~ N(0,1), amax-scaled per 1x128 / 128x128, cast to
    e4m3fn.  Generated on the device (torch casts saturate identically to the oracle for |x| <= 448)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    xa = torch.randn((m, k), device="cuda", generator=g)
    xb = torch.randn((n, k), device="cuda", generator=g)
    kb = k // 128
    sa = xa.view(m, kb, 128).abs().amax(dim=2).clamp_min(1e-30) / 448.0
    qa = (xa.view(m, kb, 128) / sa[..., None]).reshape(m, k).to(torch.float8_e4m3fn).view(torch.uint8)
    nb = n // 128
    sb = xb.view(nb, 128, kb, 128).abs().amax(dim=(1, 3)).clamp_min(1e-30) / 448.0
    qb = (xb.view(nb, 128, kb, 128) / sb[:, None, :, None]).reshape(n, k).to(torch.float8_e4m3fn).view(torch.uint8)
    return qa.contiguous(), sa.contiguous().float(), qb.contiguous(), sb.contiguous().float()


def pmc_traffic(workload: str):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (profiles/r01_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate passes and corrected as MI355X_MICROARCH.md prescribes).  PMC counters cannot be
    read from inside this process, so this is the figure of the profiled run of the same kernel, not of this run."""
    try:
        d = json.loads((ROOT / "profiles" / "r01_traffic.json").read_text())
        return int(d[workload]["traffic_bytes"])
    except Exception:
        return None


def cpu_baseline(m, n, k, a, sfa, b, sfb, budget_s=15.0):
    """The CPU oracle (oracle/dga_oracle.c, kind "port") on a bounded row sample of the same workload."""
    import numpy as np
    from oracle import oracle as O
    O.build()
    cores = max(1, min(os.cpu_count() or 1, 64))
    an = a.cpu().numpy(); bn = b.cpu().numpy(); san = sfa.cpu().numpy(); sbn = sfb.cpu().numpy()
    probe = min(m, cores)
    t0 = time.perf_counter()
    O.gemm_fp8_fp8_bf16_nt(an[:probe], san[:probe], bn, sbn, threads=cores)
    dt = max(time.perf_counter() - t0, 1e-4)
    rows = int(min(m, max(cores, probe * budget_s / dt)))
    rows -= rows % cores or 0
    rows = max(rows, cores)
    reps = 0
    t0 = time.perf_counter()
    while True:
        O.gemm_fp8_fp8_bf16_nt(an[:rows], san[:rows], bn, sbn, threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt >= 0.6 * budget_s or reps >= 50:
            break
    dt /= reps
    # the reference's own CPU path restated: np.matmul(f32, f32) per 128-wide k block (BLAS), same rows
    t1 = time.perf_counter()
    O.np_gemm_fp8_fp8_bf16_nt(an[:min(rows, 512)], san[:min(rows, 512)], bn, sbn)
    dt_blas = time.perf_counter() - t1
    return {
        "value": round(2.0 * rows * n * k / dt / 1e12, 6), "unit": "TFLOP/s", "cores": cores, "kind": "port",
        "sample": f"first {rows} of {m} rows of A against all of B, {reps} passes of {dt:.2f} s each (scalar fp32-accumulate C oracle, one thread per core)",
        "blas_value": round(2.0 * min(rows, 512) * n * k / dt_blas / 1e12, 6),
        "blas_note": "reference golden formula np.matmul(f32,f32) per k block, numpy BLAS threads",
    }


def grouped_leg(args, rank, world, dist):
    """Grouped masked-M GEMM, experts sharded over ranks (BASELINE.json configs[3]/[4])."""
    from deepgemm_ascend_amd import parallel
    return parallel.bench_grouped(rank, world, dist, steps=max(3, min(args.steps, 20)), warmup=3,
                                  groups_total=args.groups, m_max=128, n=2048, k=7168,
                                  mask=args.grouped_mask)


def _time_us(fn, iters, warm):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def widen_leg():
    """The rows either side of the hot path, reported beside the headline metric:
    the contiguous-grouped (prefill MoE) layout and the activation quantiser that feeds the GEMM."""
    import deepgemm_ascend_amd as dga
    gen = torch.Generator(device="cuda").manual_seed(7)
    groups, per, n, k = 8, 1024, 4096, 7168
    msum = groups * per
    a = _rand_fp8((msum, k), gen); b = _rand_fp8((groups, n, k), gen)
    sfa = torch.rand((msum, k // 128), device="cuda") + 0.5
    sfb = torch.rand((groups, n // 128, k // 128), device="cuda") + 0.5
    idx = torch.arange(groups, device="cuda", dtype=torch.int32).repeat_interleave(per).contiguous()
    out = torch.empty((msum, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(msum, n, k, groups=groups, contiguous=True)
    us = _time_us(lambda: dga.m_grouped_gemm_fp8_fp8_bf16_nt_contiguous((a, sfa), (b, sfb), out, idx, tiling_=t), 20, 5)
    tf = 2.0 * msum * n * k / us / 1e6
    res = {"contiguous": {"workload": f"m_grouped_gemm_fp8_fp8_bf16_nt_contiguous G={groups} x {per} rows, N={n} K={k}",
                          "tile": f"{t.m1}x{t.n1}", "kernel_us": round(us, 1),
                          "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": PEAK_FP8_TFLOPS,
                                       "unit": "TFLOP/s", "frac": round(tf / PEAK_FP8_TFLOPS, 4)}}}
    del a, b, out
    rows, kk = 32768, 7168
    x = torch.randn((rows, kk), device="cuda", dtype=torch.bfloat16)
    us = _time_us(lambda: dga.per_token_cast_to_fp8(x), 20, 5)
    byt = rows * kk * 3 + rows * (kk // 128) * 4
    res["per_token_cast"] = {"workload": f"per_token_cast_to_fp8 bf16 [{rows},{kk}] -> e4m3fn + 1x128 f32 scales",
                             "kernel_us": round(us, 1), "algorithmic_bytes": byt,
                             "roofline": {"bound": "hbm", "achieved": round(byt / us / 1e3, 1), "peak": PEAK_HBM_GBPS,
                                          "unit": "GB/s", "frac": round(byt / us / 1e3 / PEAK_HBM_GBPS, 4)}}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="dense_4096", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-grouped", action="store_true")
    ap.add_argument("--widen", action="store_true",
                    help="also time the rows either side of the hot path (contiguous-grouped layout, quantiser); off by "
                         "default so that the default command's kernel statistics hold the headline kernels only")
    ap.add_argument("--groups", type=int, default=256)
    ap.add_argument("--grouped-mask", default="full", choices=["full", "random"])
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--prewarm-ms", type=float, default=300.0,
                    help="untimed clock pre-warm in front of the W warmup steps: after idle the GPU needs a few hundred "
                         "steps to reach its sustained clocks (20 warmup steps alone leave the first 200 timed steps 14 %% "
                         "slow, scripts/warm_effect.py)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    assert world == args.gpus or world == 1, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    import deepgemm_ascend_amd as dga
    dga.lib()  # fails loudly if libdga_hip.so is missing

    m, n, k = WORKLOADS[args.workload]
    a, sfa, b, sfb = make_dense_inputs(m, n, k, seed=rank)
    out = torch.empty((m, n), dtype=torch.bfloat16, device="cuda")
    t = dga.tiling(m, n, k)

    def step():
        dga.gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out, tiling_=t)

    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.prewarm_ms:   # untimed: bring the clocks to their sustained state
        for _ in range(50):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    # HIP events on the stream the kernel is launched on (torch's current stream)
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_us = ev0.elapsed_time(ev1) * 1e3 / args.steps
    if dist:
        tt = torch.tensor([elapsed, kernel_us], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, kernel_us = float(tt[0]), float(tt[1])

    flops = 2.0 * m * n * k
    value = world * flops * args.steps / elapsed / 1e12
    achieved = flops / (kernel_us * 1e-6) / 1e12
    alg_bytes = m * k + n * k + 2 * m * n + 4 * (sfa.numel() + sfb.numel())

    grouped = None
    if not args.no_grouped:
        try:
            grouped = grouped_leg(args, rank, world, dist)
        except Exception as e:  # the primary metric must still be reported
            grouped = {"error": repr(e)}

    res = {
        "metric": "fp8 TFLOPS + % MFMA peak, 4096^3 block-scaled GEMM; grouped-GEMM tok/s at 1/2/4/8 GPU",
        "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "fp8_e4m3fn", "data": "synthetic",
        "config": {"workload": f"{args.workload}: gemm_fp8_fp8_bf16_nt M={m} N={n} K={k}, per-1x128 / per-128x128 f32 scales, bf16 out",
                   "tile": f"{t.m1}x{t.n1}x{t.k1}", "parallelism": "replicas" if world > 1 else "single",
                   "prewarm_ms": args.prewarm_ms},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_FP8_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_FP8_TFLOPS, 4),
                     "traffic": pmc_traffic("dense") if args.workload == "dense_4096" else None,
                     "kernel_us": round(kernel_us, 3), "algorithmic_bytes": alg_bytes,
                     "kernel": "gemm_fp8_blockscaled_nt_kernel"},
    }
    if grouped is not None:
        if isinstance(grouped.get("roofline"), dict) and world == 1 and args.groups == 256 and args.grouped_mask == "full":
            grouped["roofline"]["traffic"] = pmc_traffic("grouped")
        res["grouped"] = grouped
    if rank == 0 and world == 1 and args.widen:
        try:
            res["widen"] = widen_leg()
        except Exception as e:
            res["widen"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(m, n, k, a, sfa, b, sfb, budget_s=args.cpu_budget)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
