"""tests/host/sharded_host.cpp: a C++ host program (the reference's host language) that runs section 6 of INTEGRATION.md for
real -- dga_sharded_layout / dga_sharded_events_create / dga_sharded_forward with its own collective callback, no Python in
between -- at world 1 and at a world 2 emulated on one device by two host threads whose callback copies the peers' slices
ordered by events.  Every result row is compared with the CPU oracle inside the program (strict policy: byte for byte); one
case overflows an expert and reads the dropped-row counter.  This test compiles it against the in-tree libdga_hip.so and the
oracle library (the checker) and runs it."""
import os
import subprocess
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
SRC = ROOT / "tests" / "host" / "sharded_host.cpp"
OUT = ROOT / "build" / "host" / "sharded_host"


def _build(oracle):
    lib = ROOT / "deepgemm_ascend_amd" / "libdga_hip.so"
    ora = ROOT / "oracle" / "libdga_oracle.so"
    assert lib.exists(), "libdga_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    oracle.build()
    hdr = SRC.with_name("sharded_host_common.hpp")
    if OUT.exists() and OUT.stat().st_mtime >= max(SRC.stat().st_mtime, hdr.stat().st_mtime, lib.stat().st_mtime):
        return
    OUT.parent.mkdir(parents=True, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-x", "hip", f"-I{ROOT / 'include'}", str(SRC),
                           "-o", str(OUT), f"-L{lib.parent}", "-ldga_hip", f"-L{ora.parent}", "-ldga_oracle", "-lpthread",
                           f"-Wl,-rpath,{lib.parent}", f"-Wl,-rpath,{ora.parent}"])


@pytest.mark.parametrize("world", [1, 2])
def test_cpp_host_runs_the_sharded_forward(dga, oracle, world):
    _build(oracle)
    r = subprocess.run([str(OUT), str(world)], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    print(r.stdout)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "FAIL" not in r.stdout and "cases passed" in r.stdout
