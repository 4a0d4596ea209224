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

import ctypes
import time
from typing import Callable, Optional

import numpy as np
import torch


def _default_compute(a, sfa, b, sfb, out, masked_m, expected_m):
    from . import api
    api.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m)


def _default_compute_policy(a, sfa, b, sfb, out, masked_m, expected_m, policy):
    from . import api
    api.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m, policy=policy)


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
            overflow[0] += 1
            continue
        d[r] = bucket * cap + cnt[bucket]
        cnt[bucket] += 1
        if tg is not None:
            at = tag_off + int(d[r]) * tag_stride
            tg[at:at + 4] = np.frombuffer(np.int32(lo).tobytes(), np.uint8)


class ExpertShardedGroupedGemm:
    """First client of the C ABI's sharded forward (include/dga_hip.h: dga_sharded_layout / _plan / _forward).  The layout of
    the exchange buffers and the step sequence come from the library; on device tensors with the default compute the library's
    executor runs the plan (this class hands it torch.distributed.all_to_all_single as the collective callback), otherwise --
    CPU tensors of the gloo tests, an injected `compute`, per-phase timing -- the SAME plan is interpreted here."""

    def __init__(self, rank: int, world: int, groups_total: int, m_max: int, n: int, k: int, device,
                 dist=None, compute: Optional[Callable] = None, chunks: Optional[int] = None,
                 capacity_factor: Optional[float] = None, max_tokens: Optional[int] = None, strict: bool = False,
                 indexed: Optional[bool] = None, policy: Optional[str] = None, overlap: Optional[bool] = None):
        from . import _lib, api
        import os as _os
        # world > 1 on the device: overlap the exchanges with the GEMM on side streams through the library's executor.  OFF by
        # default until it has run against RCCL with more than one rank (see forward()).
        self.overlap = bool(int(_os.environ.get("DGA_SHARDED_OVERLAP", "0"))) if overlap is None else bool(overlap)
        assert groups_total % world == 0, "experts must divide evenly over ranks"
        self.rank, self.world, self.dist = rank, world, dist
        self.G = groups_total
        self.m_max, self.n, self.k = m_max, n, k
        self.device = torch.device(device)
        assert not (strict and policy not in (None, "strict")), "strict=True contradicts policy"
        self.policy = "strict" if strict else policy          # arithmetic policy of the GEMM (api.ARITHMETIC_POLICIES)
        # indexed: the grouped GEMM gathers token rows where they lie (the caller's tensors at world 1, the receive buffer
        # otherwise) and scatters result rows straight into the buffer that travels back -- no pack / unpack copy either side
        # of it.  The device default at world 1.  At world > 1 it is OPT-IN (indexed=True): there the kernel reads the receive
        # buffer of a collective through a slot table built on another stream, and that ordering has only run against an
        # in-process emulation of the exchange (tests/test_parallel_gpu.py), never against RCCL with more than one rank; the
        # packed path (one unpack copy, one gather copy) is the default until tests/test_parallel_rccl.py has run on a
        # multi-GPU box.  An injected `compute` (the CPU tests) works on the packed masked layout.
        if indexed is None:
            indexed = compute is None and self.device.type == "cuda" and world == 1
        assert not (indexed and compute is not None), "an injected compute takes the packed layout"
        # the C executor's policy word: -1 = the library's default (dga_default_policy), -2 / -3 = the fast policy's own tiling (as it is /
        # with the power-of-two-scales flag), >= 0 = a dispatchPolicyTag.  "auto" picks per SHAPE between two arithmetics and this layout
        # has one shape: name the arithmetic.
        if self.policy is not None and self.policy not in api.ARITHMETIC_POLICIES:
            raise ValueError(f"policy must be one of {sorted(api.ARITHMETIC_POLICIES)}")
        if self.policy == "auto":
            raise ValueError("policy='auto' has no meaning for the expert-sharded forward: name 'bf16_exact', 'fast', ...")
        tag = {"fast": -2, "fast_ue8m0": -3}.get(self.policy, api.ARITHMETIC_POLICIES.get(self.policy) if self.policy else None)
        self.shape = _lib.ShardedShape(world, rank, groups_total, m_max, n, k, int(chunks or 0), int(max_tokens or 0),
                                       float(capacity_factor) if capacity_factor is not None else 0.0, 1 if indexed else 0,
                                       -1 if tag is None else int(tag))
        lay = _lib.ShardedLayout()
        _lib.check(_lib.lib().dga_sharded_layout(ctypes.byref(self.shape), ctypes.byref(lay)), "sharded_layout")
        self.layout = lay
        self.Gl, self.Glc, self.chunks = lay.groups_local, lay.groups_per_chunk, lay.chunks
        self.kb, self.nb = lay.kb, lay.nb
        self.hdr, self.row_bytes = int(lay.hdr_offset), int(lay.row_bytes)
        self.indexed = bool(lay.indexed)
        self.capacity_factor = capacity_factor
        self.max_tokens = int(lay.max_tokens)
        if world > 1:
            self.C = int(lay.pair_capacity)
        steps = (_lib.ShardedStep * lay.steps)()
        cnt = ctypes.c_int(0)
        _lib.check(_lib.lib().dga_sharded_plan(ctypes.byref(self.shape), steps, lay.steps, ctypes.byref(cnt)), "sharded_plan")
        self.steps = [steps[i] for i in range(cnt.value)]
        self.compute = compute
        self._packed = None
        self.masked_m = torch.zeros((self.Gl,), dtype=torch.int32, device=device)
        self.overflow = torch.zeros((1,), dtype=torch.int32, device=device)
        self.b = None
        self.sfb = None
        self._T = -1
        self._side = None
        self._events = None
        self._cb = None
        self._cb_error = None

    def __del__(self):
        try:
            if self._events is not None:
                from . import _lib
                _lib.lib().dga_sharded_events_destroy(len(self._events), self._events)
        except Exception:
            pass

    def _packed_layout(self):
        """The masked layout [Gl, m_max, K] (+ scales, + the output): it exists only where the packed path runs."""
        if self._packed is None:
            dev = self.device
            self._packed = (torch.zeros((self.Gl, self.m_max, self.k), dtype=torch.uint8, device=dev),
                            torch.ones((self.Gl, self.m_max, self.kb), dtype=torch.float32, device=dev),
                            torch.zeros((self.Gl, self.m_max, self.n), dtype=torch.bfloat16, device=dev))
        return self._packed

    a = property(lambda self: self._packed_layout()[0])
    sfa = property(lambda self: self._packed_layout()[1])
    out = property(lambda self: self._packed_layout()[2])

    def set_weights(self, b: torch.Tensor, sfb: torch.Tensor):
        assert tuple(b.shape) == (self.Gl, self.n, self.k) and tuple(sfb.shape) == (self.Gl, self.nb, self.kb)
        assert b.is_contiguous() and sfb.is_contiguous() and sfb.dtype == torch.float32
        self.b, self.sfb = b, sfb

    def owner(self, g):
        return g // self.Gl

    def pair_capacity(self, tokens: int) -> int:
        """Rows reserved per (expert chunk, destination rank) for a rank that brings `tokens` tokens (dga_sharded_layout)."""
        from . import _lib
        sh = _lib.ShardedShape.from_buffer_copy(self.shape)
        sh.max_tokens = int(tokens)
        lay = _lib.ShardedLayout()
        _lib.check(_lib.lib().dga_sharded_layout(ctypes.byref(sh), ctypes.byref(lay)), "sharded_layout")
        return int(lay.pair_capacity) if self.world > 1 else max(1, min(int(tokens), self.Glc * self.m_max))

    def _ensure(self, tokens: int):
        """Static exchange buffers, allocated once for max_tokens rows (the same size on every rank), sized by the layout."""
        if tokens > self.max_tokens:
            raise ValueError(f"{tokens} tokens on rank {self.rank} > max_tokens {self.max_tokens} the engine was built for")
        if self._T >= 0:
            return
        dev, lay = self.device, self.layout
        self._T = self.max_tokens
        self.slot = torch.empty((max(self.max_tokens, 1),), dtype=torch.int64, device=dev)
        self.row_of_slot = torch.zeros((self.Gl * self.m_max,), dtype=torch.int64, device=dev) if self.indexed else None
        if self.world == 1:
            return
        rows = int(lay.rows_total)
        self.pair_cnt = torch.zeros((self.chunks * self.world,), dtype=torch.int32, device=dev)
        self.send = torch.zeros((rows, self.row_bytes), dtype=torch.uint8, device=dev)
        self.recv = torch.zeros((rows, self.row_bytes), dtype=torch.uint8, device=dev)
        self.rdest = torch.empty((rows,), dtype=torch.int64, device=dev)
        self.osend = torch.zeros((rows, self.n), dtype=torch.bfloat16, device=dev)
        self.oback = torch.zeros((rows, self.n), dtype=torch.bfloat16, device=dev)
        if dev.type == "cuda" and self._side is None:
            self._side = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))

    def dropped_tokens(self) -> int:
        """Synchronises; the number of token rows THIS rank dropped since the last call (a full (chunk, rank) pair slice on the
        sending side, a full expert on the receiving side: their output rows are zeros) -- one device counter that every
        routing step of forward() adds to and nothing reads inside forward().  Reading clears it."""
        n = int(self.overflow.item())
        if n:
            self.overflow.zero_()
        return n

    def check(self):
        """Synchronises and raises if a capacity was exceeded since the last check / dropped_tokens() call."""
        n = self.dropped_tokens()
        if n:
            raise ValueError(f"capacity exceeded: {n} token row(s) dropped -- an expert received more than m_max = {self.m_max} "
                             f"rows, or a (chunk, rank) pair more than its {getattr(self, 'C', self.m_max)} reserved rows")

    # ------------------------------------------------------------------ the library's executor (device tensors)
    def _forward_library(self, tok_q, tok_sf, expert_ids, expected_m, overlap: bool):
        from . import _lib, api
        L = _lib.lib()
        T = tok_q.shape[0]
        dev = self.device
        res = torch.empty((T, self.n), dtype=torch.bfloat16, device=dev)     # dropped tokens' rows: the plan's ZERO_DROPPED step
        ptr = lambda t: t.data_ptr() if t is not None else None
        bufs = _lib.ShardedBuffers()
        if self.world > 1:
            bufs.send, bufs.recv, bufs.osend, bufs.oback = ptr(self.send), ptr(self.recv), ptr(self.osend), ptr(self.oback)
            bufs.rdest, bufs.pair_cnt = ptr(self.rdest), ptr(self.pair_cnt)
        bufs.slot, bufs.row_of_slot = ptr(self.slot), ptr(self.row_of_slot)
        bufs.masked_m, bufs.overflow = ptr(self.masked_m), ptr(self.overflow)
        if not self.indexed:
            bufs.packed_a, bufs.packed_sfa, bufs.packed_out = ptr(self.a), ptr(self.sfa), ptr(self.out)
        bufs.b, bufs.sfb = ptr(self.b), ptr(self.sfb)
        with torch.cuda.device(dev):
            main = torch.cuda.current_stream(dev)
            t = api.tiling(self.m_max, self.n, self.k, groups=self.Glc, expected_m=int(expected_m or self.m_max))
            ws_ptr, ws_bytes = api._workspace(t, dev)
            bufs.workspace, bufs.workspace_bytes = ws_ptr, ws_bytes
            if self.world > 1 and overlap:
                handles = [main.cuda_stream, self._side[0].cuda_stream, self._side[1].cuda_stream]
            else:
                handles = [main.cuda_stream] * 3
            streams = (ctypes.c_void_p * 3)(*handles)
            if self.world > 1 and self._events is None:
                ev = (ctypes.c_void_p * int(self.layout.events))()
                _lib.check(L.dga_sharded_events_create(len(ev), ev), "sharded_events_create")
                self._events = ev
            if self.world > 1 and self._cb is None:
                per = int(self.layout.rows_per_chunk)

                def a2a(user, direction, chunk, send, recv, bytes_per_peer, stream):
                    try:   # the executor says which stream the collective belongs on; torch wants it as the current stream
                        sl = slice(chunk * per, (chunk + 1) * per)
                        with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev)):
                            if direction == 0:
                                self.dist.all_to_all_single(self.recv[sl], self.send[sl])
                            else:
                                self.dist.all_to_all_single(self.oback[sl], self.osend[sl])
                        return 0
                    except BaseException as e:   # never let an exception cross the C frames
                        self._cb_error = e
                        return 1
                self._cb = _lib.ALL_TO_ALL_FN(a2a)
            cb = self._cb if self.world > 1 else _lib.ALL_TO_ALL_FN()
            rc = L.dga_sharded_forward(ctypes.byref(self.shape), ctypes.byref(bufs), tok_q.data_ptr(), tok_sf.data_ptr(),
                                       expert_ids.data_ptr(), T, res.data_ptr(), int(expected_m or 0), streams,
                                       self._events if self.world > 1 else None, cb, None)
            if self._cb_error is not None:
                e, self._cb_error = self._cb_error, None
                raise e
            _lib.check(rc, "sharded_forward")
        return res

    # ------------------------------------------------------------------ the same plan, interpreted (CPU tensors, injected compute, phases)
    def _forward_interpreted(self, tok_q, tok_sf, expert_ids, expected_m, marks, overlap: bool):
        from . import _lib
        T, w, ch = tok_q.shape[0], self.world, self.chunks
        cuda = tok_q.is_cuda
        em = expected_m or self.m_max
        res = torch.empty((T, self.n), dtype=torch.bfloat16, device=self.device)
        main = torch.cuda.current_stream(self.device) if cuda else None
        side = self._side if (cuda and overlap and w > 1) else (main, main)
        streams = (main, side[0], side[1])
        events = {}
        compute = self.compute or (lambda a, sfa, b, sfb, out, mm, e: _default_compute_policy(a, sfa, b, sfb, out, mm, e, self.policy))
        if not self.indexed:
            flat_a = self.a.view(self.Gl * self.m_max, self.k)
            flat_sfa = self.sfa.view(self.Gl * self.m_max, self.kb).view(torch.uint8)
            flat_out = self.out.view(self.Gl * self.m_max, self.n).view(torch.uint8)
        phase_of = {_lib.STEP_ROUTE_SOURCE: "route", _lib.STEP_PACK: "pack", _lib.STEP_ZERO_COUNTS: "pack",
                    _lib.STEP_ROUTE_RECEIVED: "dispatch", _lib.STEP_UNPACK: "dispatch", _lib.STEP_ALL_TO_ALL_DISPATCH: "dispatch",
                    _lib.STEP_GEMM: "gemm", _lib.STEP_GATHER_OUT: "combine", _lib.STEP_ALL_TO_ALL_COMBINE: "combine",
                    _lib.STEP_RESTORE_ORDER: "unpack"}
        last_phase = None
        for st in self.steps:
            ph = phase_of.get(st.op)
            if ph is not None and last_phase is not None and ph != last_phase:
                marks(last_phase)
            last_phase = ph or last_phase
            stream = streams[st.stream]
            ctx = torch.cuda.stream(stream) if (cuda and stream is not None) else _Null()
            sl = slice(int(st.row_begin), int(st.row_begin + st.rows))
            g0, g1 = int(st.group_begin), int(st.group_begin + st.groups)
            with ctx:
                if st.op == _lib.STEP_WAIT_EVENT:
                    if cuda and overlap and w > 1:
                        stream.wait_event(events[st.event])
                elif st.op == _lib.STEP_RECORD_EVENT:
                    if cuda and overlap and w > 1:
                        events[st.event] = torch.cuda.Event(); events[st.event].record(stream)
                elif st.op == _lib.STEP_CLEAR_HEADERS:
                    self.send[:, self.hdr:self.hdr + 4] = 255
                elif st.op == _lib.STEP_ROUTE_SOURCE:
                    if w == 1:
                        _route_slots(expert_ids, 8, 0, T, self.Gl, self.m_max, self.masked_m, self.slot, self.overflow,
                                     inverse=self.row_of_slot if self.indexed else None)
                    else:
                        _route_slots(expert_ids, 8, 0, T, ch * w, self.C, self.pair_cnt, self.slot, self.overflow, key_div=self.Gl,
                                     key_sub=self.Glc, key_mul=w, tags=self.send, tag_stride=self.row_bytes, tag_off=self.hdr)
                elif st.op == _lib.STEP_PACK:
                    if w == 1:
                        _rows2(flat_a, tok_q, self.k, flat_sfa, tok_sf.view(torch.uint8), 4 * self.kb, self.slot[:T])
                    else:
                        _rows2(self.send, tok_q, self.k, self.send, tok_sf.view(torch.uint8), 4 * self.kb, self.slot[:T],
                               dst1_off=self.k)
                elif st.op == _lib.STEP_ZERO_COUNTS:
                    self.masked_m.zero_()
                elif st.op == _lib.STEP_ZERO_DROPPED:
                    res.masked_fill_((self.slot[:T] < 0)[:, None], 0)   # (no host sync) a token without a slot: nothing else writes its row
                elif st.op == _lib.STEP_ZERO_UNROUTED:
                    self.osend[sl].masked_fill_((self.rdest[sl] < 0)[:, None], 0)
                elif st.op == _lib.STEP_ALL_TO_ALL_DISPATCH:
                    self.dist.all_to_all_single(self.recv[sl], self.send[sl])
                elif st.op == _lib.STEP_ROUTE_RECEIVED:
                    _route_slots(self.recv[sl], self.row_bytes, self.hdr, int(st.rows), self.Gl, self.m_max, self.masked_m,
                                 self.rdest[sl], self.overflow, zero_counts=False,
                                 inverse=self.row_of_slot if self.indexed else None, inverse_base=int(st.row_begin))
                elif st.op == _lib.STEP_UNPACK:
                    _rows2(flat_a, self.recv[sl], self.k, flat_sfa, self.recv[sl], 4 * self.kb, self.rdest[sl], src1_off=self.k)
                elif st.op == _lib.STEP_GEMM:
                    if self.indexed:
                        from . import api
                        if w == 1:
                            api.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(tok_q, tok_sf, 0, self.kb, (self.b, self.sfb), res,
                                                                              self.row_of_slot, self.masked_m, self.m_max, em,
                                                                              policy=self.policy)
                        else:
                            api.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(
                                self.recv, self.recv, self.k, self.row_bytes // 4, (self.b[g0:g1], self.sfb[g0:g1]), self.osend,
                                self.row_of_slot[g0 * self.m_max:g1 * self.m_max], self.masked_m[g0:g1], self.m_max, em,
                                policy=self.policy)
                    else:
                        compute(self.a[g0:g1], self.sfa[g0:g1], self.b[g0:g1], self.sfb[g0:g1], self.out[g0:g1],
                                self.masked_m[g0:g1], em)
                elif st.op == _lib.STEP_GATHER_OUT:
                    _rows(self.osend[sl].view(torch.uint8), flat_out, src_index=self.rdest[sl], row_bytes=2 * self.n)
                elif st.op == _lib.STEP_ALL_TO_ALL_COMBINE:
                    self.dist.all_to_all_single(self.oback[sl], self.osend[sl])
                elif st.op == _lib.STEP_RESTORE_ORDER:
                    src = flat_out if w == 1 else self.oback.view(torch.uint8)
                    _rows(res.view(torch.uint8), src, src_index=self.slot[:T], row_bytes=2 * self.n)
                else:
                    raise ValueError(f"unknown plan step {st.op}")
        if last_phase is not None:
            marks(last_phase)
        return res

    def forward(self, tok_q, tok_sf, expert_ids, expected_m: int = 0, phase_us: Optional[dict] = None) -> torch.Tensor:
        """tok_q [T,K] u8, tok_sf [T,KB] f32, expert_ids [T] int64 (global expert of each token) -> bf16 [T,N] in token
        order.  Asynchronous on the current stream; nothing is read back (call check() to learn of a capacity overflow).
        phase_us: a dict that receives the device time of each phase (the phases then run back to back on one stream)."""
        T = tok_q.shape[0]
        assert expert_ids.dtype == torch.int64 and expert_ids.is_contiguous()
        assert tok_q.is_contiguous() and tok_sf.is_contiguous() and tok_sf.dtype == torch.float32
        assert tuple(tok_q.shape) == (T, self.k) and tuple(tok_sf.shape) == (T, self.kb)
        self._ensure(T)
        cuda = tok_q.is_cuda
        events = []

        def marks(name):
            if phase_us is not None and cuda:
                e = torch.cuda.Event(enable_timing=True); e.record(); events.append((name, e))

        # world > 1: the library's three-stream executor (dispatch stream -> GEMM on the caller's stream -> combine stream, the
        # collectives called back on ExternalStreams) has never met a real multi-rank collective -- tests/test_parallel_rccl.py
        # needs two GPUs and this pool's boxes have one.  Until that test has run once, the default at world > 1 is the SAME
        # plan interpreted here on ONE stream (no cross-stream ordering to get wrong); overlap=True (or $DGA_SHARDED_OVERLAP=1)
        # asks for the executor, and the RCCL test runs both.
        use_library = cuda and self.compute is None and phase_us is None and (self.world == 1 or self.overlap)
        if use_library:
            res = self._forward_library(tok_q, tok_sf, expert_ids, expected_m, overlap=True)
        else:
            marks("start")
            res = self._forward_interpreted(tok_q, tok_sf, expert_ids, expected_m, marks,
                                            overlap=phase_us is None and (self.overlap or not cuda))
        if phase_us is not None and cuda:
            torch.cuda.synchronize()
            for (_, e0), (name, e1) in zip(events[:-1], events[1:]):
                phase_us[name] = phase_us.get(name, 0.0) + e0.elapsed_time(e1) * 1e3
        if not cuda:
            self.check()    # CPU tensors (the gloo tests): the flag is host memory, checking costs nothing
        return res

    def run_local(self, expected_m: int = 0):
        compute = self.compute or (lambda a, sfa, b, sfb, out, mm, e: _default_compute_policy(a, sfa, b, sfb, out, mm, e, self.policy))
        compute(self.a, self.sfa, self.b, self.sfb, self.out, self.masked_m, expected_m or self.m_max)


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
        pol = eng.policy or api.default_policy()
        t = api.tiling(eng.m_max, eng.n, eng.k, groups=eng.Gl, expected_m=eng.m_max, policy=pol if pol == "bf16_exact" else None)
        name = (("gemm_fp8_bf16x_grouped_kernel" if int(t.build) == 9 else
                 "gemm_fp8_bf16x_persistent_kernel / gemm_fp8_blockscaled_nt_kernel<MATH = 1>") if pol == "bf16_exact" else
                "gemm_fp8_blockscaled_nt_persistent_kernel" if t.dispatchPolicyTag == api.POLICY_PERSISTENT
                else "gemm_fp8_blockscaled_nt_kernel")
        return {"kernel": name, "tile": f"{t.m1}x{t.n1}x{t.k1}", "dispatchPolicyTag": int(t.dispatchPolicyTag), "build": int(t.build)}
    except Exception:
        return {}


def _rand_fp8(shape, gen, device):
    """Uniformly random e4m3fn bytes without NaN codes (development scripts; the bench legs use the 8(d) recipe below)."""
    x = torch.randint(0, 256, shape, dtype=torch.uint8, device=device, generator=gen)
    return torch.where((x & 0x7F) == 0x7F, x & 0x80, x)


def _quantised_tokens(rows, k, gen, device):
    """SURVEY.md 8(d) recipe for the A operand: fp32 ~ N(0,1), amax-scaled per 1x128, cast to e4m3fn -- by the product's own
    quantiser (dga_cast_to_fp8_1x128, byte-exact against the oracle's in tests/test_cast_gpu.py)."""
    from . import api
    q = torch.empty((rows, k), dtype=torch.uint8, device=device)
    sf = torch.empty((rows, (k + 127) // 128), dtype=torch.float32, device=device)
    step = 8192
    for r0 in range(0, rows, step):
        r1 = min(rows, r0 + step)
        qq, ss = api.per_token_cast_to_fp8(torch.randn((r1 - r0, k), device=device, generator=gen))
        q[r0:r1] = qq.view(torch.uint8); sf[r0:r1] = ss
    return q, sf


def _quantised_weights(groups, n, k, gen, device):
    """... and for the B operand: per-128x128 amax scaling, expert by expert (dga_cast_to_fp8_128x128)."""
    from . import api
    assert n % 128 == 0, "the bench shapes keep an expert's rows on 128-row block boundaries"
    kb, nb = (k + 127) // 128, n // 128
    b = torch.empty((groups, n, k), dtype=torch.uint8, device=device)
    sfb = torch.empty((groups, nb, kb), dtype=torch.float32, device=device)
    step = 8
    for g0 in range(0, groups, step):
        g1 = min(groups, g0 + step)
        qq, ss = api.per_block_cast_to_fp8(torch.randn(((g1 - g0) * n, k), device=device, generator=gen))
        b[g0:g1] = qq.view(torch.uint8).view(g1 - g0, n, k); sfb[g0:g1] = ss.view(g1 - g0, nb, kb)
    return b, sfb


def masked_parity(a, sfa, b, sfb, masked, policies=("fast", "bf16_exact")) -> dict:
    """Every valid output row of a masked grouped problem under each policy against the strict kernel (which the tests pin bit
    for bit to the CPU oracle): max_ulp, frac_gt_2ulp, worst_excess_over_S with S = sum of the magnitudes of the scaled
    products (the strict kernel on |a|, |b|, |scales|).  Rows at or beyond masked_m must keep the bytes they had."""
    from . import api
    g, mmax, _ = a.shape
    n = b.shape[1]
    dev = a.device
    sentinel = -7.0
    run = lambda aa, sa, bb, sb, out, pol: api.m_grouped_gemm_fp8_fp8_bf16_nt_masked((aa, sa), (bb, sb), out, masked, mmax, policy=pol)
    exact = torch.full((g, mmax, n), sentinel, dtype=torch.bfloat16, device=dev)
    s_abs = torch.full((g, mmax, n), sentinel, dtype=torch.bfloat16, device=dev)
    run(a, sfa, b, sfb, exact, "strict")
    run(a & 0x7F, sfa.abs(), b & 0x7F, sfb.abs(), s_abs, "strict")
    valid = (torch.arange(mmax, device=dev)[None, :] < masked[:, None].to(torch.int64))[:, :, None].expand(g, mmax, n)

    def key(t):   # monotone integer map of bf16 bit patterns (+0 / -0 coincide)
        v = t.view(torch.int16).to(torch.int32)
        mag = v & 0x7FFF
        return torch.where(v < 0, -mag, mag)
    res = {"against": "strict policy on the same experts (bit-identical to the CPU oracle in tests/test_strict_gpu.py)",
           "experts": int(g), "masked_m": [int(x) for x in masked.cpu()], "elements": int(valid.sum())}
    for pol in policies:
        out = torch.full((g, mmax, n), sentinel, dtype=torch.bfloat16, device=dev)
        run(a, sfa, b, sfb, out, pol)
        torch.cuda.synchronize()
        untouched = bool((out[~valid] == sentinel).all()) if bool((~valid).any()) else True
        ulps = (key(out) - key(exact)).abs()[valid]
        f, e, ss = out.double()[valid], exact.double()[valid], s_abs.double()[valid]
        ulp = torch.exp2(torch.floor(torch.log2(e.abs().clamp_min(2.0 ** -126))) - 7)
        excess = ((f - e).abs() - 2 * ulp).clamp_min(0) / ss.clamp_min(1e-300)
        res[pol] = {"max_ulp": int(ulps.max()) if ulps.numel() else 0,
                    "frac_gt_2ulp": float((ulps > 2).double().mean()) if ulps.numel() else 0.0,
                    "worst_excess_over_S": float(excess.max()) if excess.numel() else 0.0,
                    "masked_rows_untouched": untouched}
    return res


def _gemm_only_us(eng, steps, warmup):
    for _ in range(warmup):
        eng.run_local()
    ev0 = torch.cuda.Event(enable_timing=True); ev1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        eng.run_local()
    ev1.record()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, ev0.elapsed_time(ev1) * 1e3 / steps


def _stream_roofline(eng, kernel_us):
    """Algorithmic bytes of the local grouped GEMM at the engine's current masked_m (weights of the non-empty
    experts once, the valid rows' activations, scales and outputs) against 8 TB/s."""
    rows = int(eng.masked_m.sum().item())
    active = int((eng.masked_m > 0).sum().item())
    alg = active * eng.n * eng.k + rows * (eng.k + 4 * eng.kb + 2 * eng.n) + eng.Gl * eng.nb * eng.kb * 4
    gbps = alg / (kernel_us * 1e-6) / 1e9
    return rows, {"bound": "hbm", "achieved": round(gbps, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbps / 8000.0, 4),
                  "traffic": None, "kernel_us": round(kernel_us, 2), "algorithmic_bytes": alg,
                  "tflops": round(2.0 * eng.n * eng.k * rows / (kernel_us * 1e-6) / 1e12, 1)}


def bench_grouped(rank, world, dist, steps=10, warmup=3, groups_total=256, m_max=128, n=2048, k=7168, mask="full",
                  capacity_factor=1.25, indexed=None, parity=True, policy="bf16_exact"):
    """BASELINE.json configs[3] (world 1) / configs[4] (world 8): G experts x (M<=128, K=7168, N=2048) on the SURVEY.md 8(d)
    data recipe (N(0,1), amax-quantised per 1x128 / 128x128).  Tokens are born uniformly on the ranks; `full` = every expert
    gets m_max rows, `random` = randint(0, m_max+1).  `policy` = the arithmetic of every timed figure (default: the operator's
    default, bf16-exact = inside north_star's tolerance); the GEMM-only time under the other policy is the `fast` (or
    `in_contract`) side object.  The GEMM-only figures are reported for BOTH masks, and `parity` holds the fast and the
    bf16-exact kernel against the strict one on 8 sampled experts with ragged masks."""
    dev = torch.device("cuda", torch.cuda.current_device())
    eng = ExpertShardedGroupedGemm(rank, world, groups_total, m_max, n, k, dev, dist, capacity_factor=capacity_factor,
                                   indexed=indexed, policy=policy)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    kb, nb = eng.kb, eng.nb
    eng.set_weights(*_quantised_weights(eng.Gl, n, k, g, dev))
    # tokens per expert contributed by this rank
    gc = torch.Generator().manual_seed(99)  # same on every rank
    per_expert_random = torch.randint(0, m_max + 1, (groups_total,), generator=gc)
    per_expert = torch.full((groups_total,), m_max, dtype=torch.int64) if mask == "full" else per_expert_random
    base = per_expert // world
    extra = per_expert % world
    mine = base + (rank < extra).to(torch.int64)
    expert_ids = torch.repeat_interleave(torch.arange(groups_total), mine).to(dev)
    expert_ids = expert_ids[torch.randperm(expert_ids.numel(), device=dev, generator=g)].contiguous()
    T = expert_ids.numel()
    tok_q, tok_sf = _quantised_tokens(T, k, g, dev)
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
    dropped_warm = eng.dropped_tokens()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.forward(tok_q, tok_sf, expert_ids)
    sync()
    e2e = (time.perf_counter() - t0) / steps
    # rows this rank dropped in the timed steps (a full pair slice / a full expert): read once, after the clock stopped
    dropped = eng.dropped_tokens()
    if dist is not None and world > 1:
        dd = torch.tensor([float(dropped), float(dropped_warm)], device=dev, dtype=torch.float64)
        dist.all_reduce(dd, op=dist.ReduceOp.SUM)
        dropped, dropped_warm = int(dd[0]), int(dd[1])
    # end-to-end self-check that needs no remote weights: a token's result row does not depend on where the token sits in the
    # batch (other slot, other chunk, other position in the exchange buffers), bit for bit
    forward_check = None
    try:
        r1 = eng.forward(tok_q, tok_sf, expert_ids).clone()
        perm = torch.randperm(T, device=dev, generator=g)
        r2 = eng.forward(tok_q[perm].contiguous(), tok_sf[perm].contiguous(), expert_ids[perm].contiguous())
        sync()
        eng.check()
        forward_check = {"rows": int(T), "permuted_batch_gives_the_same_rows_bitwise":
                         bool(torch.equal(r1[perm].view(torch.int16), r2.view(torch.int16))),
                         "nonzero_rows": int((r1.view(torch.int16) != 0).any(dim=1).sum())}
        del r1, r2, perm
    except Exception as e:
        forward_check = {"error": repr(e)}
    if dist is not None and world > 1:   # every rank reaches this: the line carries the AND over the ranks
        ok = torch.tensor([1.0 if forward_check.get("permuted_batch_gives_the_same_rows_bitwise") else 0.0], device=dev,
                          dtype=torch.float64)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        forward_check["on_every_rank"] = bool(float(ok[0]) == 1.0)
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
    eng.forward(tok_q, tok_sf, expert_ids, phase_us={})   # untimed: the phased form's own first-use costs (torch loads a kernel's code
    #                                                        object on its first launch: 45 ms landed in `route` before this line)
    for _ in range(3):
        for _ in range(5):
            eng.forward(tok_q, tok_sf, expert_ids)
        eng.forward(tok_q, tok_sf, expert_ids, phase_us=phases)
    phases = {kk: round(v / 3, 1) for kk, v in phases.items()}
    counts_forward = eng.masked_m.clone()      # rows every local expert received in the forward above
    # GEMM only (activations already in the masked layout on the owning rank: the same recipe, quantised in place)
    qa, qs = _quantised_tokens(eng.Gl * m_max, k, g, dev)
    eng.a.copy_(qa.view(eng.Gl, m_max, k)); eng.sfa.copy_(qs.view(eng.Gl, m_max, kb))
    del qa, qs
    gemm, kernel_us = _gemm_only_us(eng, steps, warmup)
    eng.check()
    rows_local, roof = _stream_roofline(eng, kernel_us)
    # the same GEMM under the OTHER policy: "fast" (the fp8 matrix instruction, bound by the weight stream) beside the in-contract
    # default, or the in-contract policy (bf16-exact, <= 1e-5 of the outputs beyond 2 bf16 ULP: `parity` below) beside "fast"
    other_policy = "fast" if policy == "bf16_exact" else "bf16_exact"
    gemm_x = kernel_us_x = None
    if eng.compute is None:
        keep = eng.policy
        try:
            eng.policy = other_policy
            gemm_x, kernel_us_x = _gemm_only_us(eng, max(3, steps // 2), 2)
        except Exception:
            gemm_x = kernel_us_x = None
        finally:
            eng.policy = keep
    # ... and under the OTHER mask of SURVEY.md 8(d) (full <-> randint(0, m_max + 1)), same buffers
    other = "random" if mask == "full" else "full"
    lo = rank * eng.Gl
    other_counts = (per_expert_random[lo:lo + eng.Gl] if other == "random" else torch.full((eng.Gl,), m_max)).to(torch.int32)
    eng.masked_m.copy_(other_counts.to(dev))
    gemm_o, kernel_us_o = _gemm_only_us(eng, steps, warmup)
    rows_o, roof_o = _stream_roofline(eng, kernel_us_o)
    eng.masked_m.copy_(counts_forward)
    if dist is not None and world > 1:
        tt = torch.tensor([e2e, gemm, kernel_us, gemm_o, kernel_us_o, gemm_x or 0.0, kernel_us_x or 0.0], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        e2e, gemm, kernel_us, gemm_o, kernel_us_o, gx, kx = (float(x) for x in tt)
        if gemm_x is not None:
            gemm_x, kernel_us_x = gx, kx
    phases_per_rank = None
    if dist is not None and world > 1:     # per-phase device time of every rank (the max over ranks hides a slow link)
        phases_per_rank = [None] * world
        dist.all_gather_object(phases_per_rank, phases)
    kern = _kernel_of(eng)
    roof.update(kern); roof_o.update(kern)
    res = {
        "workload": f"m_grouped_gemm_fp8_fp8_bf16_nt_masked G={groups_total} x (M<={m_max}, K={k}, N={n}), "
                    f"mask={mask}, {groups_total // world} experts/GPU",
        "data": "fp32 ~ N(0,1), amax-quantised per 1x128 (tokens) / per 128x128 (weights) to e4m3fn by the product's quantisers "
                "(SURVEY.md 8(d))",
        "policy": policy, "n_gpus": world, "tokens": total_tokens,
        "tok_per_s_gemm_only": round(total_tokens / gemm, 1),
        "tok_per_s_with_alltoall": round(total_tokens / e2e, 1),
        "ms_gemm": round(gemm * 1e3, 4), "ms_end_to_end": round(e2e * 1e3, 4),
        "ms_end_to_end_graph": round(e2e_graph * 1e3, 4) if e2e_graph else None,
        "dropped_tokens": {"timed_steps": int(dropped), "per_forward": round(dropped / max(steps, 1), 2), "warmup_steps": int(dropped_warm),
                           "note": "token rows that found a full (chunk, rank) pair slice or a full expert, summed over ranks; "
                                   "their output rows are zeros (0 = the capacity held)"},
        "forward_check": forward_check, "phase_us": phases, "phase_us_per_rank": phases_per_rank, "chunks": eng.chunks, "indexed_rows": bool(eng.indexed),
        "pair_capacity_rows": getattr(eng, "C", None), "capacity_factor": capacity_factor if world > 1 else None,
        "roofline": roof,
        ("fast" if other_policy == "fast" else "in_contract"):
            ({"policy": other_policy, "ms_gemm": round(gemm_x * 1e3, 4), "tok_per_s_gemm_only": round(total_tokens / gemm_x, 1),
              "frac_of_8TBps": round(roof["algorithmic_bytes"] / kernel_us_x / 8e6, 4),
              "tflops": round(2.0 * n * k * rows_local / kernel_us_x / 1e6, 1),
              "note": ("the same GEMM on the fp8 matrix instruction (opt-in policy: ~7e-4 of the outputs beyond 2 bf16 ULP)" if other_policy == "fast" else
                       "the grouped GEMM under the policy whose outputs stay within 2 bf16 ULP of the fp32-accumulate result on "
                       "all but <= 1e-5 of the elements (parity.bf16_exact); bound by the bf16 matrix rate at a full mask")}
             if gemm_x else None),
        f"{other}_mask": {"rows_per_gpu": rows_o, "ms_gemm": round(gemm_o * 1e3, 4),
                          "tok_per_s_gemm_only": round(rows_o * world / gemm_o, 1), "roofline": roof_o},
    }
    if parity and rank == 0:
        try:
            idx = sorted({0, 1, eng.Gl // 5, eng.Gl // 3, eng.Gl // 2, (2 * eng.Gl) // 3, eng.Gl - 2, eng.Gl - 1} & set(range(eng.Gl)))
            masks = [m_max, 1, 77, m_max - 1, 64, 16, m_max, 33][:len(idx)]
            it = torch.tensor(idx, device=dev)
            res["parity"] = masked_parity(eng.a[it].contiguous(), eng.sfa[it].contiguous(), eng.b[it].contiguous(),
                                          eng.sfb[it].contiguous(),
                                          torch.tensor([min(x, m_max) for x in masks], dtype=torch.int32, device=dev))
            res["parity"]["local_experts_sampled"] = idx
        except Exception as e:
            res["parity"] = {"error": repr(e)}
    return res
