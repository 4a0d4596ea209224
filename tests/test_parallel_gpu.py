"""GPU: the expert-sharded forward on device tensors.  One MI355X is all a gpurun box has, so
(1) world 1 runs as it is (no exchange), eager and replayed from one HIP graph;
(2) world 2 is EMULATED on the one device: two engines in two host threads, `dist` replaced by an in-process
    all_to_all_single that moves the slices between them -- every device kernel of the sharded path (header-tagged
    packing, capacity-bounded slots, receive-side counts, chunked three-stream schedule) runs exactly as it would under
    RCCL; only the wire is faked.  The compute is the strict kernel, so every token's row must equal the oracle's bits."""
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G_TOTAL, M_MAX, N, K = 8, 64, 256, 512


def _strict_compute(a, sfa, b, sfb, out, masked_m, expected_m):
    import deepgemm_ascend_amd as dga
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m, strict=True)


def _data(world, seed=3):
    rng = np.random.default_rng(seed)
    b = rng.integers(0, 120, size=(G_TOTAL, N, K), dtype=np.uint8)
    sfb = rng.uniform(0.5, 1.5, size=(G_TOTAL, N // 128, K // 128)).astype(np.float32)
    toks = []
    for r in range(world):
        T = 90 + 13 * r
        ids = rng.integers(0, G_TOTAL, size=T)
        ids[ids == 5] = 6                        # expert 5 receives nothing
        toks.append((rng.integers(0, 120, size=(T, K), dtype=np.uint8),
                     rng.uniform(0.5, 1.5, size=(T, K // 128)).astype(np.float32), ids.astype(np.int64)))
    return b, sfb, toks


def _want(oracle, q, sf, ids, b, sfb):
    want = np.zeros((len(ids), N), np.uint16)
    for g in np.unique(ids):
        rows = np.nonzero(ids == g)[0]
        want[rows] = oracle.gemm_fp8_fp8_bf16_nt(q[rows], sf[rows], b[g], sfb[g])
    return want


def _bits(t):
    return t.view(torch.int16).cpu().numpy().view(np.uint16)


def _engine(rank, world, dist, indexed, **kw):
    """indexed: the GEMM gathers / scatters rows itself (strict policy); packed: copies into the masked layout first."""
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    if indexed:
        return ExpertShardedGroupedGemm(rank, world, G_TOTAL, M_MAX, N, K, "cuda", dist, strict=True, max_tokens=128,
                                        indexed=True, **kw)
    return ExpertShardedGroupedGemm(rank, world, G_TOTAL, M_MAX, N, K, "cuda", dist, compute=_strict_compute,
                                    max_tokens=128, **kw)


@pytest.mark.parametrize("indexed", [True, False])
def test_world1_eager_and_graph(dga, oracle, indexed):
    b, sfb, toks = _data(1)
    q, sf, ids = toks[0]
    eng = _engine(0, 1, None, indexed)
    assert eng.indexed == indexed
    eng.set_weights(torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda())
    tq, tsf, tid = torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda()
    want = _want(oracle, q, sf, ids, b, sfb)
    res = eng.forward(tq, tsf, tid)
    eng.check()
    assert np.array_equal(_bits(res), want)
    assert np.array_equal(eng.masked_m.cpu().numpy(), np.bincount(ids, minlength=G_TOTAL).astype(np.int32))
    # the whole forward in one HIP graph (nothing is read back, every shape is static), replayed on fresh inputs
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        eng.forward(tq, tsf, tid)                # warm the per-stream scratch outside the capture
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = eng.forward(tq, tsf, tid)
    perm = np.random.default_rng(0).permutation(len(ids))
    tq.copy_(torch.from_numpy(q[perm]).cuda()); tsf.copy_(torch.from_numpy(sf[perm]).cuda()); tid.copy_(torch.from_numpy(ids[perm]).cuda())
    g.replay()
    torch.cuda.synchronize()
    eng.check()
    assert np.array_equal(_bits(out), want[perm])


@pytest.mark.parametrize("indexed", [True, False])
def test_tokens_beyond_an_experts_capacity_come_back_as_zero_rows(dga, oracle, indexed):
    """More tokens for expert 2 than m_max: the surplus is dropped, its result rows are zeros (written by the plan's
    ZERO_DROPPED step, not left uninitialised), every other row is right, and check() reports the overflow."""
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    b, sfb, _ = _data(1)
    rng = np.random.default_rng(11)
    T = M_MAX + 40
    ids = np.full(T, 2, np.int64); ids[::7] = 4
    q = rng.integers(0, 120, size=(T, K), dtype=np.uint8); sf = rng.uniform(0.5, 1.5, size=(T, K // 128)).astype(np.float32)
    eng = ExpertShardedGroupedGemm(0, 1, G_TOTAL, M_MAX, N, K, "cuda", None, strict=True, max_tokens=T, indexed=indexed)
    eng.set_weights(torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda())
    junk = torch.full((T, N), 7.0, dtype=torch.bfloat16, device="cuda"); del junk     # the allocator hands dirty memory back
    res = eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda())
    torch.cuda.synchronize()
    got = _bits(res)
    slot = eng.slot[:T].cpu().numpy()
    dropped = slot < 0
    assert dropped.sum() == (ids == 2).sum() - M_MAX and (ids[dropped] == 2).all()
    assert (got[dropped] == 0).all()
    want = _want(oracle, q, sf, ids, b, sfb)
    assert np.array_equal(got[~dropped], want[~dropped])
    # the device counter says how many rows were dropped (read and cleared by dropped_tokens(); check() raises on it)
    assert eng.dropped_tokens() == int(dropped.sum())
    assert eng.dropped_tokens() == 0
    eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda())
    with pytest.raises(ValueError, match=f"{int(dropped.sum())} token row"):
        eng.check()


def test_phase_timings_are_reported(dga):
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    b, sfb, toks = _data(1)
    q, sf, ids = toks[0]
    eng = ExpertShardedGroupedGemm(0, 1, G_TOTAL, M_MAX, N, K, "cuda", None, max_tokens=128)
    eng.set_weights(torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda())
    ph = {}
    eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda(), phase_us=ph)
    assert set(ph) == {"route", "gemm"} and all(v >= 0 for v in ph.values())     # indexed: no pack / unpack phase


class _FakeDist:
    """all_to_all_single between engines that live in threads of one process: equal splits, device copies ORDERED BY EVENTS
    on the stream the engine calls it on -- no device synchronisation anywhere, so the dispatch / GEMM / combine streams of
    an engine (and the two engines) really run concurrently, as they would around an RCCL collective.  The host barrier only
    makes sure both ranks have posted their buffers."""

    def __init__(self, rank, world, box, barrier):
        self.rank, self.world, self.box, self.barrier = rank, world, box, barrier

    def all_to_all_single(self, out, inp):
        s = torch.cuda.current_stream()
        ready = torch.cuda.Event(); ready.record(s)                 # my send slice is complete at this point of MY stream
        self.box[self.rank] = (inp, ready, None)
        self.barrier.wait()
        per = inp.shape[0] // self.world
        for src in range(self.world):
            buf, ev, _ = self.box[src]
            s.wait_event(ev)                                        # the peer's slice is complete
            out[src * per:(src + 1) * per].copy_(buf[self.rank * per:(self.rank + 1) * per], non_blocking=True)
        done = torch.cuda.Event(); done.record(s)                   # I have read every peer's buffer
        self.box[self.rank] = (inp, ready, done)
        self.barrier.wait()
        for src in range(self.world):                               # nobody overwrites a send buffer a peer is still reading
            s.wait_event(self.box[src][2])
        self.barrier.wait()


def _library_packed_engine(rank, world, dist, **kw):
    """The packed path through the LIBRARY's executor (no injected compute): strict policy, unpack / gather copies."""
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    return ExpertShardedGroupedGemm(rank, world, G_TOTAL, M_MAX, N, K, "cuda", dist, strict=True, max_tokens=128,
                                    indexed=False, **kw)


@pytest.mark.parametrize("chunks,capacity_factor,indexed", [(1, None, True), (2, None, True), (2, 2.5, True),
                                                            (2, None, False), (2, None, "library_packed"),
                                                            (1, 2.5, "library_packed")])
def test_world2_emulated_on_one_device(dga, oracle, chunks, capacity_factor, indexed):
    """indexed True / "library_packed": dga_sharded_forward (the C ABI's executor) runs the plan, this test's fake exchange is
    its collective callback; indexed False: the same plan interpreted in Python around an injected compute."""
    world = 2
    b, sfb, toks = _data(world)
    gl = G_TOTAL // world
    box, barrier = [None] * world, threading.Barrier(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            mk = _library_packed_engine if indexed == "library_packed" else (lambda r, w, d, **kw: _engine(r, w, d, indexed, **kw))
            eng = mk(rank, world, _FakeDist(rank, world, box, barrier), chunks=chunks, capacity_factor=capacity_factor)
            assert eng.indexed == (indexed is True)
            eng.set_weights(torch.from_numpy(b[rank * gl:(rank + 1) * gl]).cuda(),
                            torch.from_numpy(sfb[rank * gl:(rank + 1) * gl]).cuda())
            q, sf, ids = toks[rank]
            for _ in range(2):                   # twice: the second forward reuses every static buffer
                res = eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda())
            torch.cuda.synchronize()
            eng.check()
            results[rank] = (_bits(res), eng.masked_m.cpu().numpy().copy())
        except Exception as e:                   # a dead thread must not leave its peer at the barrier
            errors.append(e)
            barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]; [t.join(timeout=120) for t in ts]
    assert not errors, errors
    total = np.zeros(G_TOTAL, np.int64)
    for r in range(world):
        q, sf, ids = toks[r]
        total += np.bincount(ids, minlength=G_TOTAL)
        assert np.array_equal(results[r][0], _want(oracle, q, sf, ids, b, sfb)), f"rank {r}"
    assert np.array_equal(np.concatenate([results[0][1], results[1][1]]), total.astype(np.int32))


@pytest.mark.parametrize("strict", [False, True])
def test_indexed_gemm_equals_packed_gemm(dga, strict):
    """dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed on payload rows (fp8 bytes, scales and a header in one row,
    rows in scrambled order, BASELINE configs[3] K and N) writes, bit for bit, what the packed masked call computes on
    the same rows copied into the [G, m_max, K] layout; rows it does not own stay untouched."""
    g, m_max, n, k = 6, 128, 2048, 7168
    kb = k // 128
    row_bytes = k + 4 * kb + 16
    rng = np.random.default_rng(7)
    masked = np.array([128, 0, 37, 128, 1, 100], np.int32)
    rows = int(masked.sum()) + 11                                    # 11 rows belong to nobody
    payload = rng.integers(0, 120, size=(rows, row_bytes), dtype=np.uint8)
    sf = rng.uniform(0.5, 1.5, size=(rows, kb)).astype(np.float32)
    payload[:, k:k + 4 * kb] = sf.view(np.uint8)
    b = rng.integers(0, 120, size=(g, n, k), dtype=np.uint8)
    sfb = rng.uniform(0.5, 1.5, size=(g, n // 128, kb)).astype(np.float32)
    perm = rng.permutation(rows)
    row_index = np.full((g, m_max), -1, np.int64)
    at = 0
    for i in range(g):
        row_index[i, :masked[i]] = perm[at:at + masked[i]]
        at += masked[i]
    a_packed = np.zeros((g, m_max, k), np.uint8); sfa_packed = np.ones((g, m_max, kb), np.float32)
    for i in range(g):
        a_packed[i, :masked[i]] = payload[row_index[i, :masked[i]], :k]
        sfa_packed[i, :masked[i]] = sf[row_index[i, :masked[i]]]
    dev = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    tb, tsfb, tm = dev(b), dev(sfb), dev(masked)
    out_packed = torch.zeros((g, m_max, n), dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((dev(a_packed), dev(sfa_packed)), (tb, tsfb), out_packed, tm, 64, strict=strict)
    tp = dev(payload)
    out_rows = torch.full((rows, n), -5.0, dtype=torch.bfloat16, device="cuda")
    dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(tp, tp, k, row_bytes // 4, (tb, tsfb), out_rows, dev(row_index.reshape(-1)),
                                                      tm, m_max, 64, strict=strict, sync=True)
    got, want = _bits(out_rows), _bits(out_packed)
    owned = np.zeros(rows, bool)
    for i in range(g):
        idx = row_index[i, :masked[i]]
        owned[idx] = True
        assert np.array_equal(got[idx], want[i, :masked[i]]), f"group {i}"
    assert (got[~owned] == _bits(torch.tensor([-5.0], dtype=torch.bfloat16))[0]).all()


def test_config4_full_size_forward_properties(dga):
    """BASELINE configs[3] at full size through the sharded forward at world 1 (256 experts x (M <= 128, K = 7168, N = 2048), 32 768
    tokens on the SURVEY 8(d) recipe), checked by properties that need no CPU reference of that size: (a) a permuted batch gives the
    same rows bit for bit, through the library path and through the HIP graph; (b) the rows of four sampled experts equal the masked
    grouped GEMM called directly on those experts' tokens (strict policy on both sides: bit-identical)."""
    from deepgemm_ascend_amd import parallel
    dev = torch.device("cuda", 0)
    g_total, m_max, n, k = 256, 128, 2048, 7168
    gen = torch.Generator(device=dev).manual_seed(11)
    eng = parallel.ExpertShardedGroupedGemm(0, 1, g_total, m_max, n, k, dev, None, policy="strict")
    b, sfb = parallel._quantised_weights(g_total, n, k, gen, dev)
    eng.set_weights(b, sfb)
    per = torch.randint(64, m_max + 1, (g_total,), generator=torch.Generator().manual_seed(5))
    ids = torch.repeat_interleave(torch.arange(g_total), per).to(dev)
    ids = ids[torch.randperm(ids.numel(), device=dev, generator=gen)].contiguous()
    T = ids.numel()
    q, sf = parallel._quantised_tokens(T, k, gen, dev)
    r1 = eng.forward(q, sf, ids).clone()
    eng.check()
    perm = torch.randperm(T, device=dev, generator=gen)
    r2 = eng.forward(q[perm].contiguous(), sf[perm].contiguous(), ids[perm].contiguous()).clone()
    eng.check()
    assert torch.equal(r1[perm].view(torch.int16), r2.view(torch.int16))
    assert int((r1.view(torch.int16) != 0).any(dim=1).sum()) == T
    for e in (0, 77, 200, 255):
        rows = torch.nonzero(ids == e).flatten()
        mm = rows.numel()
        a = torch.zeros((1, m_max, k), dtype=torch.uint8, device=dev); a[0, :mm] = q.view(torch.uint8)[rows]
        sa = torch.ones((1, m_max, k // 128), dtype=torch.float32, device=dev); sa[0, :mm] = sf[rows]
        out = torch.zeros((1, m_max, n), dtype=torch.bfloat16, device=dev)
        dga.m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sa), (b[e:e + 1].contiguous(), sfb[e:e + 1].contiguous()), out,
                                                  torch.tensor([mm], dtype=torch.int32, device=dev), mm, strict=True, sync=True)
        assert torch.equal(out[0, :mm].view(torch.int16), r1[rows].view(torch.int16)), e
