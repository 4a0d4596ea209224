"""Expert sharding of the grouped masked-M GEMM over the GPUs of one node.

The reference has no collective of any kind ("multi-card" = rank-sliced independent processes,
/root/reference/deep_gemm_ascend/benchmark_msprof/main.cpp:24-26,
framework/benchmark/benchmark.py:249-253); this module is new work for BASELINE.json configs[4].

Partitioning: expert g lives on rank g // (G / world); its weights never move.  One exchange each way:
  dispatch  all-to-all of token rows (fp8 [K] bytes + fp32 [K/128] scales + a 4-byte header naming the expert on its
            owner, packed in one byte row) from the token's home rank to the expert's rank -> masked layout [G_local, m_max, K]
  compute   m_grouped_gemm_fp8_fp8_bf16_nt_masked with masked_m = rows received per expert (counted on the device)
  combine   all-to-all of bf16 [N] rows back, restored to the original token order

No host round trip: every shape is static.  A rank sends each peer a fixed-capacity slice (`pair capacity` rows per
(expert chunk, destination rank); unused rows carry header -1) and the per-expert row counts are built on the
receiving device by atomics (dga_route_slots), so nothing is read back and a whole forward captures into one HIP graph.
The price is padding on the wire: capacity = capacity_factor x the even share (default: the provable bound, which
never overflows before an expert itself does); a bucket that does overflow raises a sticky device flag that
`check()` turns into an error.  The experts are processed in `chunks`: dispatch of chunk i+1, the grouped GEMM of
chunk i and the combine of chunk i-1 run on three streams.  `torch.distributed` backend "nccl" is RCCL on ROCm; on
MI355X's fully connected xGMI mesh every peer pair has its own link, so an all-to-all is per-link bound.
world == 1 has no exchange: tokens are routed straight into the masked layout (one indexed copy in, one out).

`compute` is injectable so that the routing can be covered by world_size-2 gloo tests on CPU with the oracle
as the checker; the default is the HIP operator (no CPU fallback in the product path).
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import numpy as np
import torch


def _default_compute(a, sfa, b, sfb, out, masked_m, expected_m):
    from . import api
    api.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m)


def _default_compute_strict(a, sfa, b, sfb, out, masked_m, expected_m, strict):
    from . import api
    api.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m, strict=strict)


def _rows(dst, src, dst_index=None, src_index=None, row_bytes=None, dst_off=0, src_off=0):
    """Indexed row copy on byte views: the HIP kernel (dga_copy_rows) for device tensors; torch indexing for the CPU
    tensors of the gloo routing tests (where nothing in this module touches a GPU).  Negative indices are skipped."""
    if dst.is_cuda:
        from . import api
        api.copy_rows(dst, src, dst_index, src_index, row_bytes=row_bytes, dst_byte_offset=dst_off,
                      src_byte_offset=src_off)
        return
    rb = row_bytes if row_bytes is not None else min(dst.shape[1], src.shape[1])
    n = dst_index.numel() if dst_index is not None else (src_index.numel() if src_index is not None else src.shape[0])
    di = dst_index if dst_index is not None else torch.arange(n)
    si = src_index if src_index is not None else torch.arange(n)
    ok = (di >= 0) & (si >= 0)
    dst[di[ok], dst_off:dst_off + rb] = src[si[ok], src_off:src_off + rb]


def _rows2(dst0, src0, bytes0, dst1, src1, bytes1, dst_index, dst1_off=0, src1_off=0):
    """Two row streams with one index list (fp8 bytes and scales of a token): dga_copy_rows2 on the device."""
    if dst0.is_cuda:
        from . import api
        api.copy_rows2(dst0, src0, bytes0, dst1, src1, bytes1, dst_index=dst_index, dst1_off=dst1_off, src1_off=src1_off)
        return
    _rows(dst0, src0, dst_index=dst_index, row_bytes=bytes0)
    _rows(dst1, src1, dst_index=dst_index, row_bytes=bytes1, dst_off=dst1_off, src_off=src1_off)


def _route_slots(keys, key_stride, key_off, rows, buckets, cap, counts, dest, overflow, key_div=1, key_sub=0, key_mul=1,
                 zero_counts=True, tags=None, tag_stride=0, tag_off=0, inverse=None, inverse_base=0):
    """dga_route_slots on the device; the same assignment in numpy for the CPU tensors of the gloo tests."""
    if dest.is_cuda:
        from . import api
        api.route_slots(keys, key_stride, rows, buckets, cap, counts, dest, overflow, key_div=key_div, key_sub=key_sub,
                        key_mul=key_mul, zero_counts=zero_counts, tags=tags, tag_stride_bytes=tag_stride,
                        keys_byte_offset=key_off, tags_byte_offset=tag_off, inverse=inverse, inverse_base=inverse_base)
        return
    assert inverse is None, "the indexed GEMM is a device path"
    kb = keys.contiguous().view(torch.uint8).reshape(-1).numpy()
    key = np.array([np.frombuffer(kb[key_off + r * key_stride: key_off + r * key_stride + 4].tobytes(), np.int32)[0]
                    for r in range(rows)], np.int64)
    cnt = counts.numpy()
    if zero_counts:
        cnt[:buckets] = 0
    d = dest.numpy()
    tg = tags.view(torch.uint8).reshape(-1).numpy() if tags is not None else None
    for r in range(rows):
        d[r] = -1
        if key[r] < 0:
            continue
        hi, lo = divmod(int(key[r]), key_div)
        bucket = (lo // key_sub) * key_mul + hi if key_sub else hi
        if hi >= (key_mul if key_sub else buckets) or bucket >= buckets:
            continue
        if cnt[bucket] >= cap:
            overflow[0] = 1
            continue
        d[r] = bucket * cap + cnt[bucket]
        cnt[bucket] += 1
        if tg is not None:
            at = tag_off + int(d[r]) * tag_stride
            tg[at:at + 4] = np.frombuffer(np.int32(lo).tobytes(), np.uint8)


class ExpertShardedGroupedGemm:
    def __init__(self, rank: int, world: int, groups_total: int, m_max: int, n: int, k: int, device,
                 dist=None, compute: Optional[Callable] = None, chunks: Optional[int] = None,
                 capacity_factor: Optional[float] = None, max_tokens: Optional[int] = None, strict: bool = False,
                 indexed: Optional[bool] = None):
        assert groups_total % world == 0, "experts must divide evenly over ranks"
        self.rank, self.world, self.dist = rank, world, dist
        self.G, self.Gl = groups_total, groups_total // world
        self.m_max, self.n, self.k = m_max, n, k
        self.kb = (k + 127) // 128
        self.nb = (n + 127) // 128
        self.hdr = k + 4 * self.kb                      # byte offset of the 4-byte header in a payload row
        # rows of the exchange buffers start on 128-byte lines: the indexed GEMM reads a row's k blocks (128 B each) where they
        # lie, and a row stride that is not a line multiple would make every such read straddle two lines
        self.row_bytes = (self.hdr + 4 + 127) // 128 * 128
        self.device = torch.device(device)
        self.strict = strict
        # indexed (device default): the grouped GEMM gathers token rows where they lie (the caller's tensors at world 1,
        # the receive buffer otherwise) and scatters result rows straight into the buffer that travels back -- no
        # pack / unpack copy either side of it.  An injected `compute` (the CPU tests) works on the packed masked layout.
        self.indexed = (compute is None and self.device.type == "cuda") if indexed is None else bool(indexed)
        assert not (self.indexed and compute is not None), "an injected compute takes the packed layout"
        self.compute = compute or (lambda a, sfa, b, sfb, out, mm, em: _default_compute_strict(a, sfa, b, sfb, out, mm, em, strict))
        if chunks is None:
            chunks = 2 if (world > 1 and self.Gl % 2 == 0 and self.Gl >= 8) else 1
        assert self.Gl % chunks == 0, "chunks must divide the experts per rank"
        self.chunks, self.Glc = chunks, self.Gl // chunks
        self.capacity_factor = capacity_factor
        # the largest number of tokens one rank brings to a forward: it sizes the exchange slices, so every rank must
        # pass the same value (default: what a rank's own experts can hold)
        self.max_tokens = int(max_tokens) if max_tokens is not None else self.Gl * m_max
        # resident buffers sized once (288 GB HBM: weights + masked activations stay put)
        self.a = torch.zeros((self.Gl, m_max, k), dtype=torch.uint8, device=device)
        self.sfa = torch.ones((self.Gl, m_max, self.kb), dtype=torch.float32, device=device)
        self.out = torch.zeros((self.Gl, m_max, n), dtype=torch.bfloat16, device=device)
        self.masked_m = torch.zeros((self.Gl,), dtype=torch.int32, device=device)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=device)
        self.b = None
        self.sfb = None
        self._T = -1
        self._side = None

    def set_weights(self, b: torch.Tensor, sfb: torch.Tensor):
        assert tuple(b.shape) == (self.Gl, self.n, self.k) and tuple(sfb.shape) == (self.Gl, self.nb, self.kb)
        self.b, self.sfb = b, sfb

    def owner(self, g):
        return g // self.Gl

    def pair_capacity(self, tokens: int) -> int:
        """Rows reserved per (expert chunk, destination rank).  The bound min(tokens, experts in the chunk x m_max) cannot
        overflow before an expert does; capacity_factor trades that guarantee for less padding on the wire."""
        bound = max(1, min(tokens, self.Glc * self.m_max))
        if self.capacity_factor is None:
            return bound
        even = tokens / float(self.world * self.chunks)
        return int(min(bound, max(16, -(-int(np.ceil(self.capacity_factor * even)) // 16) * 16)))

    def _ensure(self, tokens: int):
        """Static exchange buffers, allocated once for max_tokens rows (the same size on every rank)."""
        if tokens > self.max_tokens:
            raise ValueError(f"{tokens} tokens on rank {self.rank} > max_tokens {self.max_tokens} the engine was built for")
        if self._T >= 0:
            return
        dev, w, ch = self.device, self.world, self.chunks
        self._T = self.max_tokens
        self.slot = torch.empty((max(self.max_tokens, 1),), dtype=torch.int64, device=dev)
        self.row_of_slot = torch.zeros((self.Gl * self.m_max,), dtype=torch.int64, device=dev) if self.indexed else None
        if w == 1:
            self.indexed = self.indexed and self.max_tokens * self.k < 2 ** 31 - 1    # 32-bit offsets in the tile loads
            return
        C = self.C = self.pair_capacity(self.max_tokens)
        rows = ch * w * C
        self.pair_cnt = torch.zeros((ch * w,), dtype=torch.int32, device=dev)
        self.send = torch.zeros((rows, self.row_bytes), dtype=torch.uint8, device=dev)
        self.recv = torch.zeros((rows, self.row_bytes), dtype=torch.uint8, device=dev)
        self.rdest = torch.empty((rows,), dtype=torch.int64, device=dev)
        self.osend = torch.zeros((rows, self.n), dtype=torch.bfloat16, device=dev)
        self.oback = torch.zeros((rows, self.n), dtype=torch.bfloat16, device=dev)
        self.indexed = self.indexed and rows * self.row_bytes < 2 ** 31 - 1
        if dev.type == "cuda" and self._side is None:
            self._side = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))

    def check(self):
        """Synchronises and raises if a capacity was exceeded since the last check (rows of a full bucket are dropped)."""
        if int(self.overflow.item()):
            self.overflow.zero_()
            raise ValueError(f"capacity exceeded: an expert received more than m_max = {self.m_max} rows, or a "
                             f"(chunk, rank) pair more than its {getattr(self, 'C', self.m_max)} reserved rows")

    # ------------------------------------------------------------------ world 1: route straight into the masked layout
    def _forward_local(self, tok_q, tok_sf, expert_ids, expected_m, marks):
        T = tok_q.shape[0]
        if self.indexed:
            from . import api
            _route_slots(expert_ids, 8, 0, T, self.Gl, self.m_max, self.masked_m, self.slot, self.overflow,
                         inverse=self.row_of_slot)
            marks("route")
            res = torch.empty((T, self.n), dtype=self.out.dtype, device=self.out.device)
            api.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(tok_q, tok_sf, 0, self.kb, (self.b, self.sfb), res,
                                                              self.row_of_slot, self.masked_m, self.m_max,
                                                              expected_m or self.m_max, strict=self.strict)
            marks("gemm")
            return res
        flat_a = self.a.view(self.Gl * self.m_max, self.k)
        flat_sfa = self.sfa.view(self.Gl * self.m_max, self.kb).view(torch.uint8)
        _route_slots(expert_ids, 8, 0, T, self.Gl, self.m_max, self.masked_m, self.slot, self.overflow)
        marks("route")
        _rows2(flat_a, tok_q, self.k, flat_sfa, tok_sf.view(torch.uint8), 4 * self.kb, self.slot[:T])
        marks("pack")
        self.compute(self.a, self.sfa, self.b, self.sfb, self.out, self.masked_m, expected_m or self.m_max)
        marks("gemm")
        res = torch.empty((T, self.n), dtype=self.out.dtype, device=self.out.device)
        _rows(res.view(torch.uint8), self.out.view(self.Gl * self.m_max, self.n).view(torch.uint8), src_index=self.slot[:T],
              row_bytes=2 * self.n)
        marks("unpack")
        return res

    # ------------------------------------------------------------------ world > 1
    def _forward_sharded(self, tok_q, tok_sf, expert_ids, expected_m, marks, overlap):
        T, w, ch, C = tok_q.shape[0], self.world, self.chunks, self.C
        per = w * C                                             # rows of one chunk's exchange
        flat_a = self.a.view(self.Gl * self.m_max, self.k)
        flat_sfa = self.sfa.view(self.Gl * self.m_max, self.kb).view(torch.uint8)
        flat_out = self.out.view(self.Gl * self.m_max, self.n).view(torch.uint8)
        cuda = tok_q.is_cuda
        # ---- source side: slot of every token in its (chunk, destination) slice, header = expert index on its owner
        self.send[:, self.hdr:self.hdr + 4] = 255              # every header -1: rows nobody fills are skipped by the receiver
        _route_slots(expert_ids, 8, 0, T, ch * w, C, self.pair_cnt, self.slot, self.overflow, key_div=self.Gl,
                     key_sub=self.Glc, key_mul=w, tags=self.send, tag_stride=self.row_bytes, tag_off=self.hdr)
        marks("route")
        _rows2(self.send, tok_q, self.k, self.send, tok_sf.view(torch.uint8), 4 * self.kb, self.slot[:T], dst1_off=self.k)
        self.masked_m.zero_()
        marks("pack")
        main = torch.cuda.current_stream(self.device) if cuda else None
        s_disp, s_comb = self._side if (cuda and overlap) else (main, main)
        ev_d, ev_g, ev_c = [], [], []

        def on(stream):
            return torch.cuda.stream(stream) if cuda else _Null()

        if cuda and overlap:
            s_disp.wait_stream(main)
        for c in range(ch):                                     # dispatch: exchange, then receive-side slots + scatter
            with on(s_disp):
                sl = slice(c * per, (c + 1) * per)
                self.dist.all_to_all_single(self.recv[sl], self.send[sl])
                _route_slots(self.recv[sl], self.row_bytes, self.hdr, per, self.Gl, self.m_max, self.masked_m,
                             self.rdest[sl], self.overflow, zero_counts=False, inverse=self.row_of_slot,
                             inverse_base=c * per)
                if not self.indexed:
                    _rows2(flat_a, self.recv[sl], self.k, flat_sfa, self.recv[sl], 4 * self.kb, self.rdest[sl],
                           src1_off=self.k)
                if cuda and overlap:
                    ev_d.append(torch.cuda.Event()); ev_d[-1].record(s_disp)
        marks("dispatch")
        for c in range(ch):                                     # grouped GEMM of the chunk's experts
            if cuda and overlap:
                main.wait_event(ev_d[c])
            g0, g1 = c * self.Glc, (c + 1) * self.Glc
            if self.indexed:    # rows read from the receive buffer, results written into the buffer that travels back
                from . import api
                api.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(
                    self.recv, self.recv, self.k, self.row_bytes // 4, (self.b[g0:g1], self.sfb[g0:g1]), self.osend,
                    self.row_of_slot[g0 * self.m_max:g1 * self.m_max], self.masked_m[g0:g1], self.m_max,
                    expected_m or self.m_max, strict=self.strict)
            else:
                self.compute(self.a[g0:g1], self.sfa[g0:g1], self.b[g0:g1], self.sfb[g0:g1], self.out[g0:g1],
                             self.masked_m[g0:g1], expected_m or self.m_max)
            if cuda and overlap:
                ev_g.append(torch.cuda.Event()); ev_g[-1].record(main)
        marks("gemm")
        for c in range(ch):                                     # combine: rows back in arrival order, exchange
            with on(s_comb):
                if cuda and overlap:
                    s_comb.wait_event(ev_g[c])
                sl = slice(c * per, (c + 1) * per)
                if not self.indexed:
                    _rows(self.osend[sl].view(torch.uint8), flat_out, src_index=self.rdest[sl], row_bytes=2 * self.n)
                self.dist.all_to_all_single(self.oback[sl], self.osend[sl])
                if cuda and overlap:
                    ev_c.append(torch.cuda.Event()); ev_c[-1].record(s_comb)
        if cuda and overlap:
            for e in ev_c:
                main.wait_event(e)
        marks("combine")
        res = torch.empty((T, self.n), dtype=self.out.dtype, device=self.out.device)
        _rows(res.view(torch.uint8), self.oback.view(torch.uint8), src_index=self.slot[:T], row_bytes=2 * self.n)
        marks("unpack")
        return res

    def forward(self, tok_q, tok_sf, expert_ids, expected_m: int = 0, phase_us: Optional[dict] = None) -> torch.Tensor:
        """tok_q [T,K] u8, tok_sf [T,KB] f32, expert_ids [T] int64 (global expert of each token) -> bf16 [T,N] in token
        order.  Asynchronous on the current stream; nothing is read back (call check() to learn of a capacity overflow).
        phase_us: a dict that receives the device time of each phase (the phases then run back to back on one stream)."""
        T = tok_q.shape[0]
        assert expert_ids.dtype == torch.int64 and expert_ids.is_contiguous()
        self._ensure(T)
        cuda = tok_q.is_cuda
        events = []

        def marks(name):
            if phase_us is not None and cuda:
                e = torch.cuda.Event(enable_timing=True); e.record(); events.append((name, e))

        marks("start")
        if self.world == 1:
            res = self._forward_local(tok_q, tok_sf, expert_ids, expected_m, marks)
        else:
            res = self._forward_sharded(tok_q, tok_sf, expert_ids, expected_m, marks, overlap=phase_us is None)
        if phase_us is not None and cuda:
            torch.cuda.synchronize()
            for (_, e0), (name, e1) in zip(events[:-1], events[1:]):
                phase_us[name] = phase_us.get(name, 0.0) + e0.elapsed_time(e1) * 1e3
        if not cuda:
            self.check()    # CPU tensors (the gloo tests): the flag is host memory, checking costs nothing
        return res

    def run_local(self, expected_m: int = 0):
        self.compute(self.a, self.sfa, self.b, self.sfb, self.out, self.masked_m, expected_m or self.m_max)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


# ---------------------------------------------------------------------- benchmark leg (called from bench.py)

def _kernel_of(eng) -> dict:
    """Name and tile of the kernel the local grouped GEMM runs (for the roofline object; the rocprofv3 summary lists the
    kernel under this name)."""
    try:
        from . import api
        t = api.tiling(eng.m_max, eng.n, eng.k, groups=eng.Gl, expected_m=eng.m_max)
        name = ("gemm_fp8_blockscaled_nt_persistent_kernel" if t.dispatchPolicyTag == api.POLICY_PERSISTENT
                else "gemm_fp8_blockscaled_nt_kernel")
        return {"kernel": name, "tile": f"{t.m1}x{t.n1}x{t.k1}", "dispatchPolicyTag": int(t.dispatchPolicyTag)}
    except Exception:
        return {}


def _rand_fp8(shape, gen, device):
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device=device, generator=gen)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)


def bench_grouped(rank, world, dist, steps=10, warmup=3, groups_total=256, m_max=128, n=2048, k=7168, mask="full",
                  capacity_factor=1.25):
    """BASELINE.json configs[3] (world 1) / configs[4] (world 8): G experts x (M<=128, K=7168, N=2048).
    Tokens are born uniformly on the ranks; `full` = every expert gets m_max rows, `random` = randint(0, m_max+1)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = ExpertShardedGroupedGemm(rank, world, groups_total, m_max, n, k, dev, dist, capacity_factor=capacity_factor)
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
    expert_ids = expert_ids[torch.randperm(expert_ids.numel(), device=dev, generator=g)].contiguous()
    T = expert_ids.numel()
    tok_q = _rand_fp8((T, k), g, dev)
    tok_sf = torch.rand((T, kb), device=dev, generator=g) + 0.5
    total_tokens = int(per_expert.sum())

    def sync():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # end to end: dispatch + GEMM + combine, eager launches
    for _ in range(warmup):
        eng.forward(tok_q, tok_sf, expert_ids)
    sync()
    eng.check()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(tok_q, tok_sf, expert_ids)
    sync()
    e2e = (time.perf_counter() - t0) / steps
    # the same forward replayed from one HIP graph (possible because no phase reads anything back)
    e2e_graph = None
    if world == 1:
        try:
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr):
                eng.forward(tok_q, tok_sf, expert_ids)
            for _ in range(warmup):
                gr.replay()
            sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                gr.replay()
            sync()
            e2e_graph = (time.perf_counter() - t0) / steps
        except Exception:   # capture is an optimisation, never a requirement of the bench line
            e2e_graph = None
    # per-phase device time (phases back to back on one stream)
    # (each phased forward follows five ordinary ones on the same stream without a sync in between: measured after an idle
    #  gap the GEMM phase would show the clock ramp -- up to +20 % -- instead of what it costs inside the timed loop)
    phases = {}
    for _ in range(3):
        for _ in range(5):
            eng.forward(tok_q, tok_sf, expert_ids)
        eng.forward(tok_q, tok_sf, expert_ids, phase_us=phases)
    phases = {kk: round(v / 3, 1) for kk, v in phases.items()}
    # GEMM only (activations already in the masked layout on the owning rank; random bytes there -- the indexed forward
    # never fills that layout, and zeros would run at a higher clock than real data)
    eng.a.copy_(_rand_fp8(tuple(eng.a.shape), g, dev))
    eng.sfa.copy_(torch.rand(tuple(eng.sfa.shape), device=dev, generator=g) + 0.5)
    for _ in range(warmup):
        eng.run_local()
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
    eng.check()
    if dist is not None and world > 1:
        tt = torch.tensor([e2e, gemm, kernel_us], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e2e, gemm, kernel_us = (float(x) for x in tt)
    rows_local = int(eng.masked_m.sum().item())
    active = int((eng.masked_m > 0).sum().item())
    alg_bytes = active * n * k + rows_local * (k + 4 * kb + 2 * n) + eng.Gl * nb * kb * 4
    flops_local = 2.0 * n * k * rows_local
    res = {
        "workload": f"m_grouped_gemm_fp8_fp8_bf16_nt_masked G={groups_total} x (M<={m_max}, K={k}, N={n}), "
                    f"mask={mask}, {groups_total // world} experts/GPU",
        "n_gpus": world, "tokens": total_tokens,
        "tok_per_s_gemm_only": round(total_tokens / gemm, 1),
        "tok_per_s_with_alltoall": round(total_tokens / e2e, 1),
        "ms_gemm": round(gemm * 1e3, 4), "ms_end_to_end": round(e2e * 1e3, 4),
        "ms_end_to_end_graph": round(e2e_graph * 1e3, 4) if e2e_graph else None,
        "phase_us": phases, "chunks": eng.chunks, "indexed_rows": bool(eng.indexed),
        "pair_capacity_rows": getattr(eng, "C", None), "capacity_factor": capacity_factor if world > 1 else None,
        "roofline": {"bound": "hbm", "achieved": round(alg_bytes / (kernel_us * 1e-6) / 1e9, 1), "peak": 8000.0,
                     "unit": "GB/s", "frac": round(alg_bytes / (kernel_us * 1e-6) / 1e9 / 8000.0, 4),
                     "traffic": None, "kernel_us": round(kernel_us, 2), "algorithmic_bytes": alg_bytes,
                     "tflops": round(flops_local / (kernel_us * 1e-6) / 1e12, 1), **_kernel_of(eng)},
    }
    return res
