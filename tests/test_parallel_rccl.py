"""GPU, >= 2 devices: the sharded forward under the REAL collective (RCCL through torch.distributed "nccl"), world 2, one
process per GPU.  Skipped on a one-GPU box (the gpurun boxes): there the exchange is emulated in-process
(tests/test_parallel_gpu.py).  What it checks when it can run: the default single-stream forward AND the library's three-stream
executor (overlap=True), each with the indexed rows (GEMM reading the receive buffer in place) and with the packed layout, one
and two chunks -- all equal the oracle's rows under the strict policy byte for byte, twice in a row (static buffers reused).
This is the evidence parallel.py waits for before `overlap` and `indexed` become the defaults at world > 1."""
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

WORKER = r'''
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["DGA_ROOT"])
from oracle import oracle as O
from deepgemm_ascend_amd.parallel import ExpertShardedGroupedGemm

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(rank)
dist.init_process_group("nccl", device_id=torch.device("cuda", rank))
G, MMAX, N, K = 8, 64, 256, 512
rng = np.random.default_rng(3)
b = rng.integers(0, 120, size=(G, N, K), dtype=np.uint8)
sfb = rng.uniform(0.5, 1.5, size=(G, N // 128, K // 128)).astype(np.float32)
toks = []
for r in range(world):
    T = 90 + 13 * r
    ids = rng.integers(0, G, size=T); ids[ids == 5] = 6
    toks.append((rng.integers(0, 120, size=(T, K), dtype=np.uint8), rng.uniform(0.5, 1.5, size=(T, K // 128)).astype(np.float32),
                 ids.astype(np.int64)))
q, sf, ids = toks[rank]
gl = G // world
outs = {}
# overlap False = the world > 1 default (the plan interpreted on one stream); True = the library's three-stream executor with
# the collectives called back on its streams (opt-in until this very test has run on a two-GPU box)
for overlap in (False, True):
    for indexed in (True, False):
        for chunks in (1, 2):
            eng = ExpertShardedGroupedGemm(rank, world, G, MMAX, N, K, "cuda", dist, strict=True, max_tokens=128, indexed=indexed,
                                           chunks=chunks, overlap=overlap)
            eng.set_weights(torch.from_numpy(b[rank * gl:(rank + 1) * gl]).cuda(), torch.from_numpy(sfb[rank * gl:(rank + 1) * gl]).cuda())
            for _ in range(2):
                res = eng.forward(torch.from_numpy(q).cuda(), torch.from_numpy(sf).cuda(), torch.from_numpy(ids).cuda())
            torch.cuda.synchronize()
            assert eng.dropped_tokens() == 0
            outs[(overlap, indexed, chunks)] = res.view(torch.int16).cpu().numpy().view(np.uint16)
want = np.zeros((len(ids), N), np.uint16)
for g in np.unique(ids):
    rows = np.nonzero(ids == g)[0]
    want[rows] = O.gemm_fp8_fp8_bf16_nt(q[rows], sf[rows], b[g], sfb[g])
for key, got in outs.items():
    assert np.array_equal(got, want), f"rank {rank} {key}: differs from the oracle"
dist.barrier()
dist.destroy_process_group()
print(f"rank {rank} ok")
'''


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL with more than one rank)")
def test_world2_rccl_indexed_equals_packed_equals_oracle(dga, oracle, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, DGA_ROOT=str(ROOT), HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "rank 0 ok" in r.stdout and "rank 1 ok" in r.stdout
