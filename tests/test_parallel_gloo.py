"""CPU, world_size 2 over gloo: the expert-sharded grouped path (dispatch all-to-all -> local grouped GEMM ->
combine all-to-all).  The HIP operator cannot run here, so `compute` is injected with the CPU oracle -- the
oracle stays the checker: what is under test is the routing / exchange / masked-layout logic of
deepgemm_ascend_amd.parallel, compared against a single-process evaluation of the same tokens."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
G_TOTAL, M_MAX, N, K = 4, 32, 128, 256


def _oracle_compute(a, sfa, b, sfb, out, masked_m, expected_m):
    from oracle import oracle as O
    init = out.view(torch.int16).numpy().view(np.uint16)
    res = O.m_grouped_gemm_fp8_fp8_bf16_nt_masked(a.numpy(), sfa.numpy(), b.numpy(), sfb.numpy(), init, masked_m.numpy())
    out.copy_(torch.from_numpy(res.view(np.int16)).view(torch.bfloat16))


def _world_data():
    rng = np.random.default_rng(11)
    b = rng.integers(0, 120, size=(G_TOTAL, N, K), dtype=np.uint8)
    sfb = rng.uniform(0.5, 1.5, size=(G_TOTAL, 1, 2)).astype(np.float32)
    toks = []
    for r in range(2):
        T = 21 + 5 * r
        ids = rng.integers(0, G_TOTAL, size=T)
        ids[ids == 2] = 3                       # expert 2 receives nothing: an empty expert
        q = rng.integers(0, 120, size=(T, K), dtype=np.uint8)
        sf = rng.uniform(0.5, 1.5, size=(T, 2)).astype(np.float32)
        toks.append((q, sf, ids))
    return b, sfb, toks


def _worker(rank, world, port, ret, chunks=1, capacity_factor=None):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    b, sfb, toks = _world_data()
    eng = ExpertShardedGroupedGemm(rank, world, G_TOTAL, M_MAX, N, K, torch.device("cpu"), dist, compute=_oracle_compute,
                                   chunks=chunks, capacity_factor=capacity_factor, max_tokens=32)
    gl = G_TOTAL // world
    eng.set_weights(torch.from_numpy(b[rank * gl:(rank + 1) * gl]), torch.from_numpy(sfb[rank * gl:(rank + 1) * gl]))
    q, sf, ids = toks[rank]
    res = eng.forward(torch.from_numpy(q), torch.from_numpy(sf), torch.from_numpy(ids))
    ret[rank] = (res.view(torch.int16).numpy().view(np.uint16).copy(), eng.masked_m.numpy().copy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("chunks,capacity_factor", [(1, None), (2, None), (2, 3.0)])
def test_expert_sharded_grouped_gemm_world2(chunks, capacity_factor):
    """Static-shape exchange (fixed-capacity slices, header-routed rows, device-side counts), whole and in two expert
    chunks, with the provable capacity and with a capacity factor: every token's row bit-equal to the oracle."""
    from oracle import oracle as O
    O.build()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret, chunks, capacity_factor), nprocs=2, join=True)
    b, sfb, toks = _world_data()
    total = np.zeros(G_TOTAL, np.int64)
    for r in range(2):
        q, sf, ids = toks[r]
        total += np.bincount(ids, minlength=G_TOTAL)
        # every token's row = its own 1 x K row against its expert's weights
        want = np.zeros((len(ids), N), np.uint16)
        for t in range(len(ids)):
            g = ids[t]
            want[t] = O.gemm_fp8_fp8_bf16_nt(q[t:t + 1], sf[t:t + 1], b[g], sfb[g])[0]
        got, masked = ret[r]
        assert np.array_equal(got, want), f"rank {r}"
    assert np.array_equal(np.concatenate([ret[0][1], ret[1][1]]), total.astype(np.int32))
    assert total[2] == 0


def test_world1_is_exchange_free():
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    from oracle import oracle as O
    b, sfb, toks = _world_data()
    eng = ExpertShardedGroupedGemm(0, 1, G_TOTAL, M_MAX, N, K, torch.device("cpu"), None, compute=_oracle_compute)
    eng.set_weights(torch.from_numpy(b), torch.from_numpy(sfb))
    q, sf, ids = toks[0]
    res = eng.forward(torch.from_numpy(q), torch.from_numpy(sf), torch.from_numpy(ids))
    got = res.view(torch.int16).numpy().view(np.uint16)
    for t in range(len(ids)):
        assert np.array_equal(got[t], O.gemm_fp8_fp8_bf16_nt(q[t:t + 1], sf[t:t + 1], b[ids[t]], sfb[ids[t]])[0])


def test_capacity_overflow_is_an_error():
    """Five tokens for an expert with room for four: the device flag is raised, check() (forward() on CPU tensors) throws."""
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    eng = ExpertShardedGroupedGemm(0, 1, 2, 4, N, K, torch.device("cpu"), None, compute=lambda *a: None)
    eng.set_weights(torch.zeros((2, N, K), dtype=torch.uint8), torch.ones((2, 1, 2)))
    with pytest.raises(ValueError):
        eng.forward(torch.zeros((5, K), dtype=torch.uint8), torch.ones((5, 2)), torch.zeros(5, dtype=torch.int64))
    assert int(eng.masked_m[0]) == 4            # the rows that fitted are in place, masked_m never exceeds m_max
    eng.forward(torch.zeros((4, K), dtype=torch.uint8), torch.ones((4, 2)), torch.zeros(4, dtype=torch.int64))  # flag was cleared


def test_pair_capacity_rule():
    from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm
    mk = lambda **kw: ExpertShardedGroupedGemm(0, 8, 256, 128, N, K, torch.device("cpu"), None, compute=lambda *a: None, **kw)
    assert mk(chunks=1).pair_capacity(4096) == 4096               # provable bound: min(T, experts x m_max)
    assert mk(chunks=2).pair_capacity(4096) == 2048
    assert mk(chunks=2, capacity_factor=1.25).pair_capacity(4096) == 320   # 1.25 x 4096 / 16 buckets
    assert mk(chunks=1, capacity_factor=1.25).pair_capacity(4096) == 640
    assert mk(chunks=1, capacity_factor=100.0).pair_capacity(4096) == 4096
