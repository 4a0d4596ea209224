"""Expert sharding of the grouped masked-M GEMM over the GPUs of one node.

The reference has no collective of any kind ("multi-card" = rank-sliced independent processes,
/root/reference/deep_gemm_ascend/benchmark_msprof/main.cpp:24-26,
framework/benchmark/benchmark.py:249-253); this module is new work for BASELINE.json configs[4].

Partitioning: expert g lives on rank g // (G / world); its weights never move.  One exchange each way:
  dispatch  all-to-all-v of token rows (fp8 [K] bytes + their fp32 [K/128] scales packed in one byte row)
            from the token's home rank to the expert's rank -> masked layout [G_local, m_max, K]
  compute   m_grouped_gemm_fp8_fp8_bf16_nt_masked with masked_m = received counts
  combine   all-to-all-v of bf16 [N] rows back, restored to the original token order
`torch.distributed` backend "nccl" is RCCL on ROCm; on MI355X's fully connected xGMI mesh every peer pair has
its own link, so an all-to-all is per-link bound.  world == 1 skips the exchange.

`compute` is injectable so that the routing can be covered by world_size-2 gloo tests on CPU with the oracle
as the checker; the default is the HIP operator (no CPU fallback in the product path).
"""
from __future__ import annotations

import time
from dataclasses import dataclass
from typing import Callable, Optional

import numpy as np
import torch


@dataclass
class RouteState:
    order: torch.Tensor        # scatter=False: permutation that sorts my tokens by expert; scatter=True: slot of every token
    dest: torch.Tensor         # slot (g_local * m_max + row) of every received row
    send_splits: list
    recv_splits: list
    tokens: int
    scatter: bool = False      # device path: `order` is pos (token -> slot in expert-sorted order, dga_route_tokens)


def _default_compute(a, sfa, b, sfb, out, masked_m, expected_m):
    from . import api
    api.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m)


def _rows(dst, src, dst_index=None, src_index=None, row_bytes=None, dst_off=0, src_off=0):
    """Indexed row copy on byte views: the HIP kernel (dga_copy_rows) for device tensors; torch indexing for the CPU
    tensors of the gloo routing tests (where nothing in this module touches a GPU)."""
    if dst.is_cuda:
        from . import api
        api.copy_rows(dst, src, dst_index, src_index, row_bytes=row_bytes, dst_byte_offset=dst_off,
                      src_byte_offset=src_off)
        return
    rb = row_bytes if row_bytes is not None else min(dst.shape[1], src.shape[1])
    s = src[:, src_off:src_off + rb] if src_index is None else src[src_index, src_off:src_off + rb]
    if dst_index is None:
        dst[:s.shape[0], dst_off:dst_off + rb] = s
    else:
        dst[dst_index, dst_off:dst_off + rb] = s


class ExpertShardedGroupedGemm:
    def __init__(self, rank: int, world: int, groups_total: int, m_max: int, n: int, k: int, device,
                 dist=None, compute: Optional[Callable] = None):
        assert groups_total % world == 0, "experts must divide evenly over ranks"
        self.rank, self.world, self.dist = rank, world, dist
        self.G, self.Gl = groups_total, groups_total // world
        self.m_max, self.n, self.k = m_max, n, k
        self.kb = (k + 127) // 128
        self.nb = (n + 127) // 128
        self.row_bytes = k + 4 * self.kb
        self.device = device
        self.compute = compute or _default_compute
        # resident buffers sized once (288 GB HBM: weights + masked activations stay put)
        self.a = torch.zeros((self.Gl, m_max, k), dtype=torch.uint8, device=device)
        self.sfa = torch.ones((self.Gl, m_max, self.kb), dtype=torch.float32, device=device)
        self.out = torch.zeros((self.Gl, m_max, n), dtype=torch.bfloat16, device=device)
        self.masked_m = torch.zeros((self.Gl,), dtype=torch.int32, device=device)
        self.b = None
        self.sfb = None
        self._pinned = None
        self._stage = [None, None]
        self._stage_i = 0

    def set_weights(self, b: torch.Tensor, sfb: torch.Tensor):
        assert tuple(b.shape) == (self.Gl, self.n, self.k) and tuple(sfb.shape) == (self.Gl, self.nb, self.kb)
        self.b, self.sfb = b, sfb

    def owner(self, g):
        return g // self.Gl

    # ------------------------------------------------------------------ dispatch
    def dispatch(self, tok_q: torch.Tensor, tok_sf: torch.Tensor, expert_ids: torch.Tensor) -> RouteState:
        """tok_q [T,K] u8, tok_sf [T,KB] f32, expert_ids [T] int64 (global expert of each token)."""
        T = tok_q.shape[0]
        scatter = tok_q.is_cuda
        if scatter:
            # one pass of atomics: counts and the slot of every token in the expert-sorted order (no device sort, no
            # histogram with its hidden host sync)
            from . import api
            counts, order = api.route_tokens(expert_ids.contiguous(), self.G)
        else:
            counts = torch.bincount(expert_ids, minlength=self.G).to(torch.int64)        # [G]
            order = None
        if self.world > 1:
            flat = torch.empty((self.world * self.G,), dtype=torch.int64, device=counts.device)
            self.dist.all_gather_into_tensor(flat, counts)
        else:
            flat = counts
        # The one host synchronisation of the exchange: the count matrix (world x G int64) comes to the host, where
        # the split lists that all_to_all_single needs anyway, the per-expert row counts and the slot of every
        # arriving row are derived with numpy (a few microseconds) instead of a dozen small device launches.  The
        # copy is asynchronous into pinned memory; the sort and the packing below do not depend on it and run meanwhile.
        if flat.is_cuda:
            if self._pinned is None or self._pinned.numel() != flat.numel():
                self._pinned = torch.empty((flat.numel(),), dtype=torch.int64, pin_memory=True)
            self._pinned.copy_(flat, non_blocking=True)
            landed = torch.cuda.Event()
            landed.record()
        else:
            landed = None
        # one byte row per token: K fp8 bytes followed by KB fp32 scales, in expert order
        payload = torch.empty((T, self.row_bytes), dtype=torch.uint8, device=tok_q.device)
        if scatter:
            from . import api
            api.copy_rows2(payload, tok_q, self.k, payload, tok_sf.view(torch.uint8), 4 * self.kb, dst_index=order,
                           dst1_off=self.k)
        else:
            order = torch.argsort(expert_ids, stable=True).contiguous()
            _rows(payload, tok_q, src_index=order, row_bytes=self.k)
            _rows(payload, tok_sf.view(torch.uint8), src_index=order, row_bytes=4 * self.kb, dst_off=self.k)
        if landed is not None:
            landed.synchronize()
            allc = self._pinned.numpy().reshape(self.world, self.G).copy()
        else:
            allc = flat.numpy().reshape(self.world, self.G)
        mine = allc[:, self.rank * self.Gl:(self.rank + 1) * self.Gl]                 # [world, Gl] rows I receive
        send_splits = allc[self.rank].reshape(self.world, self.Gl).sum(1).tolist()
        recv_splits = mine.sum(1).tolist()
        masked = mine.sum(0)                                                          # [Gl]
        if masked.size and int(masked.max()) > self.m_max:
            raise ValueError(f"an expert received {int(masked.max())} rows > m_max {self.m_max}")
        total = int(sum(recv_splits))
        # slot of every received row: source-major, expert-minor arrival order -> [g, row] masked layout
        start = (np.cumsum(mine, 0) - mine) + (np.arange(self.Gl, dtype=np.int64) * self.m_max)[None, :]
        flat_cnt = mine.reshape(-1)
        seg_begin = np.cumsum(flat_cnt) - flat_cnt
        dest_np = np.repeat(start.reshape(-1) - seg_begin, flat_cnt) + np.arange(total, dtype=np.int64)
        if tok_q.is_cuda:
            # staged through pinned memory so that the two uploads are asynchronous (a pageable source makes the copy a
            # blocking staging copy); a fresh buffer pair per call while the previous one may still be in flight
            slot = self._stage[self._stage_i]
            self._stage_i ^= 1
            if slot is None or slot[0].numel() < total:
                slot = (torch.empty((max(total, 1024),), dtype=torch.int64, pin_memory=True),
                        torch.empty((self.Gl,), dtype=torch.int32, pin_memory=True), torch.cuda.Event())
                self._stage[self._stage_i ^ 1] = slot
            else:
                slot[2].synchronize()   # the uploads that used this slot two calls ago
            slot[0][:total].numpy()[:] = dest_np
            slot[1].numpy()[:] = masked
            dest = slot[0][:total].to(tok_q.device, non_blocking=True)
            self.masked_m.copy_(slot[1], non_blocking=True)
            slot[2].record()
        else:
            dest = torch.from_numpy(dest_np)
            self.masked_m.copy_(torch.from_numpy(masked.astype(np.int32)))
        if self.world > 1:
            recv = torch.empty((total, self.row_bytes), dtype=torch.uint8, device=tok_q.device)
            self.dist.all_to_all_single(recv, payload, recv_splits, send_splits)
        else:
            recv = payload
        if scatter:
            api.copy_rows2(self.a.view(self.Gl * self.m_max, self.k), recv, self.k,
                           self.sfa.view(self.Gl * self.m_max, self.kb).view(torch.uint8), recv, 4 * self.kb,
                           dst_index=dest, src1_off=self.k)
        else:
            _rows(self.a.view(self.Gl * self.m_max, self.k), recv, dst_index=dest, row_bytes=self.k)
            _rows(self.sfa.view(self.Gl * self.m_max, self.kb).view(torch.uint8), recv, dst_index=dest,
                  row_bytes=4 * self.kb, src_off=self.k)
        return RouteState(order, dest, send_splits, recv_splits, T, scatter)

    # ------------------------------------------------------------------ compute
    def run_local(self, expected_m: int = 0):
        self.compute(self.a, self.sfa, self.b, self.sfb, self.out, self.masked_m, expected_m or self.m_max)

    # ------------------------------------------------------------------ combine
    def combine(self, st: RouteState) -> torch.Tensor:
        total = st.dest.numel()
        rows = torch.empty((total, self.n), dtype=self.out.dtype, device=self.out.device)   # arrival order
        _rows(rows.view(torch.uint8), self.out.view(self.Gl * self.m_max, self.n).view(torch.uint8), src_index=st.dest,
              row_bytes=2 * self.n)
        if self.world > 1:
            back = torch.empty((st.tokens, self.n), dtype=self.out.dtype, device=rows.device)
            self.dist.all_to_all_single(back, rows, st.send_splits, st.recv_splits)
        else:
            back = rows
        res = torch.empty_like(back)
        if st.scatter:
            _rows(res.view(torch.uint8), back.view(torch.uint8), src_index=st.order, row_bytes=2 * self.n)
        else:
            _rows(res.view(torch.uint8), back.view(torch.uint8), dst_index=st.order, row_bytes=2 * self.n)
        return res

    def forward(self, tok_q, tok_sf, expert_ids, expected_m: int = 0) -> torch.Tensor:
        st = self.dispatch(tok_q, tok_sf, expert_ids)
        self.run_local(expected_m)
        return self.combine(st)


# ---------------------------------------------------------------------- benchmark leg (called from bench.py)

def _rand_fp8(shape, gen, device):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device=device, generator=gen)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)


def bench_grouped(rank, world, dist, steps=10, warmup=3, groups_total=256, m_max=128, n=2048, k=7168, mask="full"):
    """BASELINE.json configs[3] (world 1) / configs[4] (world 8): G experts x (M<=128, K=7168, N=2048).
    Tokens are born uniformly on the ranks; `full` = every expert gets m_max rows, `random` = randint(0, m_max+1)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = ExpertShardedGroupedGemm(rank, world, groups_total, m_max, n, k, dev, dist)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    kb, nb = eng.kb, eng.nb
    eng.set_weights(_rand_fp8((eng.Gl, n, k), g, dev), torch.rand((eng.Gl, nb, kb), device=dev, generator=g) + 0.5)
    # tokens per expert contributed by this rank
    gc = torch.Generator().manual_seed(99)  # same on every rank
    if mask == "full":
        per_expert = torch.full((groups_total,), m_max, dtype=torch.int64)
    else:
        per_expert = torch.randint(0, m_max + 1, (groups_total,), generator=gc)
    base = per_expert // world
    extra = per_expert % world
    mine = base + (rank < extra).to(torch.int64)
    expert_ids = torch.repeat_interleave(torch.arange(groups_total), mine).to(dev)
    expert_ids = expert_ids[torch.randperm(expert_ids.numel(), device=dev, generator=g)]
    T = expert_ids.numel()
    tok_q = _rand_fp8((T, k), g, dev)
    tok_sf = torch.rand((T, kb), device=dev, generator=g) + 0.5
    total_tokens = int(per_expert.sum())

    def sync():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # end to end: dispatch + GEMM + combine
    for _ in range(warmup):
        eng.forward(tok_q, tok_sf, expert_ids)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(tok_q, tok_sf, expert_ids)
    sync()
    e2e = (time.perf_counter() - t0) / steps
    # GEMM only (activations already in the masked layout on the owning rank)
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    sync()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        eng.run_local()
    ev1.record()
    sync()
    gemm = (time.perf_counter() - t0) / steps
    kernel_us = ev0.elapsed_time(ev1) * 1e3 / steps
    if dist is not None and world > 1:
        tt = torch.tensor([e2e, gemm, kernel_us], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e2e, gemm, kernel_us = (float(x) for x in tt)
    rows_local = int(eng.masked_m.sum().item())
    active = int((eng.masked_m > 0).sum().item())
    alg_bytes = active * n * k + rows_local * (k + 4 * kb + 2 * n) + eng.Gl * nb * kb * 4
    flops_local = 2.0 * n * k * rows_local
    return {
        "workload": f"m_grouped_gemm_fp8_fp8_bf16_nt_masked G={groups_total} x (M<={m_max}, K={k}, N={n}), "
                    f"mask={mask}, {groups_total // world} experts/GPU",
        "n_gpus": world, "tokens": total_tokens,
        "tok_per_s_gemm_only": round(total_tokens / gemm, 1),
        "tok_per_s_with_alltoall": round(total_tokens / e2e, 1),
        "ms_gemm": round(gemm * 1e3, 4), "ms_end_to_end": round(e2e * 1e3, 4),
        "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (kernel_us * 1e-6) / 1e9, 1), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(alg_bytes / (kernel_us * 1e-6) / 1e9 / 8000.0, 4),
                     "traffic": None, "kernel_us": round(kernel_us, 2), "algorithmic_bytes": alg_bytes,
                     "tflops": round(flops_local / (kernel_us * 1e-6) / 1e12, 1)},
    }
