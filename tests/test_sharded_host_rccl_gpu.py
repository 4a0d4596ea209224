"""tests/host/sharded_host_rccl.cpp: the C++ host of the expert-sharded forward over a REAL RCCL communicator -- one process per GPU,
ncclCommInitRank, the collective callback of INTEGRATION.md section 6 (ncclGroupStart / ncclSend + ncclRecv per peer / ncclGroupEnd on
the stream the executor names) -- every result row against the CPU oracle inside the program.  Needs two GPUs: skipped on the
one-GPU boxes (the link check runs on the CPU box: tests/test_build.py::test_the_rccl_host_compiles_and_links).  The ranks are fresh
child processes (this process has initialised the GPU: it starts them and waits, it does not exec)."""
import os
import subprocess
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "host" / "sharded_host_rccl.cpp"
OUT = ROOT / "build" / "host" / "sharded_host_rccl"


def _build(oracle):
    lib = ROOT / "deepgemm_ascend_amd" / "libdga_hip.so"
    ora = ROOT / "oracle" / "libdga_oracle.so"
    assert lib.exists(), "libdga_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    oracle.build()
    hdr = SRC.with_name("sharded_host_common.hpp")
    if OUT.exists() and OUT.stat().st_mtime >= max(SRC.stat().st_mtime, hdr.stat().st_mtime, lib.stat().st_mtime):
        return
    OUT.parent.mkdir(parents=True, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", f"-I{ROOT / 'include'}",
                           "-I/opt/rocm/include", str(SRC), "-o", str(OUT), f"-L{lib.parent}", "-ldga_hip", f"-L{ora.parent}", "-ldga_oracle",
                           "-L/opt/rocm/lib", "-lrccl", "-lpthread", f"-Wl,-rpath,{lib.parent}", f"-Wl,-rpath,{ora.parent}",
                           "-Wl,-rpath,/opt/rocm/lib"])


@pytest.mark.parametrize("world", [2])
def test_cpp_host_runs_the_sharded_forward_over_rccl(dga, oracle, world, tmp_path):
    if torch.cuda.device_count() < world:
        pytest.skip(f"{torch.cuda.device_count()} GPU(s) visible, {world} needed")
    _build(oracle)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    id_file = tmp_path / "rccl_id.bin"
    procs = [subprocess.Popen([str(OUT), str(world), str(r), str(id_file)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env)
             for r in range(world)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        print(o)
        assert p.returncode == 0, f"rank {r}: exit {p.returncode}\n{o[-3000:]}"
        assert "FAIL" not in o and "cases passed" in o
