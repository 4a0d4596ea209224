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

G_TOTAL, M_MAX, N, K = 8, 32, 256, 512


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


def test_world1_eager_and_graph(dga, oracle):
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    b, sfb, toks = _data(1)
    q, sf, ids = toks[0]
    eng = ExpertShardedGroupedGemm(0, 1, G_TOTAL, M_MAX, N, K, "cuda", None, compute=_strict_compute, max_tokens=128)
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


def test_phase_timings_are_reported(dga):
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    b, sfb, toks = _data(1)
    q, sf, ids = toks[0]
    eng = ExpertShardedGroupedGemm(0, 1, G_TOTAL, M_MAX, N, K, "cuda", None, max_tokens=128)
    eng.set_weights(torch.from_numpy(b).cuda(), torch.from_numpy(sfb).cuda())
    ph = {}
    eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda(), phase_us=ph)
    assert set(ph) == {"route", "pack", "gemm", "unpack"} and all(v >= 0 for v in ph.values())


class _FakeDist:
    """all_to_all_single between engines that live in threads of one process (equal splits, device copies)."""

    def __init__(self, rank, world, box, barrier):
        self.rank, self.world, self.box, self.barrier = rank, world, box, barrier

    def all_to_all_single(self, out, inp):
        self.box[self.rank] = inp
        torch.cuda.synchronize()
        self.barrier.wait()
        per = inp.shape[0] // self.world
        for src in range(self.world):
            out[src * per:(src + 1) * per].copy_(self.box[src][self.rank * per:(self.rank + 1) * per])
        torch.cuda.synchronize()
        self.barrier.wait()


@pytest.mark.parametrize("chunks,capacity_factor", [(1, None), (2, None), (2, 2.5)])
def test_world2_emulated_on_one_device(dga, oracle, chunks, capacity_factor):
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    world = 2
    b, sfb, toks = _data(world)
    gl = G_TOTAL // world
    box, barrier = [None] * world, threading.Barrier(world)
    results, errors = [None] * world, []

    def run(rank):
        try:
            torch.cuda.set_device(0)
            eng = ExpertShardedGroupedGemm(rank, world, G_TOTAL, M_MAX, N, K, "cuda", _FakeDist(rank, world, box, barrier),
                                           compute=_strict_compute, chunks=chunks, capacity_factor=capacity_factor,
                                           max_tokens=128)
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
