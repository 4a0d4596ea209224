"""GPU: the bbit harness binary end to end: gen files -> run -> verify, both file formats."""
import os
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent
BBIT = ROOT / "deepgemm_ascend_amd" / "dga_kernels_bbit"


def _run(tmp_path, args):
    env = dict(os.environ, DGA_BBIT_ITERS="3", DGA_BBIT_WARMUP="1")
    r = subprocess.run([str(BBIT)] + [str(a) for a in args], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    return r


@pytest.mark.parametrize("mode,m,n,k", [("fp8", 192, 320, 400), ("fp16", 64, 96, 80)])
def test_bbit_roundtrip(dga, tmp_path, monkeypatch, mode, m, n, k):
    from deepgemm_ascend_amd.harness import files
    monkeypatch.chdir(tmp_path)
    files.gen_golden_data(m, n, k, mode=mode, seed=1)
    r = _run(tmp_path, [0, m, n, k, 1, 1, 3, 8, 20, 10])
    assert r.returncode == 0, r.stderr
    assert re.search(r"Task Duration\(us\): (\d+\.\d+)", r.stdout)      # what benchmark.py:411 parses
    assert files.verify_result("output/output.bin", "output/golden.bin", mode=mode if mode == "fp8" else "fp16",
                               rtol=None if mode == "fp8" else 2e-5)


def test_bbit_argument_errors(dga, tmp_path):
    assert _run(tmp_path, [0, 16, 16, 16]).returncode == 2                   # argc != 11 (benchmark_util.h:50)
    assert _run(tmp_path, [0, 16, 16, 16, 1, 1, 0, 8, 20, 10]).returncode == 2  # zero knob
    assert _run(tmp_path, [0, 16, 16, 16, 1, 1, 3, 8, 20, 10]).returncode == 3  # missing input files
