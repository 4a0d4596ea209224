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


def _run(tmp_path, args, **extra_env):
    env = dict(os.environ, DGA_BBIT_ITERS="3", DGA_BBIT_WARMUP="1", **extra_env)
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


def test_bbit_verifier_is_the_one_parity_bar(dga, tmp_path, monkeypatch):
    """The file verifier states the product's single bar (harness/tolerance.py): the fast path's output passes at 2^-15 S,
    the bf16-exact policy's ($DGA_BF16_EXACT=1) also at ITS bar (2^-22 S, 1e-5 of the elements), and an output file with
    one element pushed past 2 ulp + eps * S is rejected -- the old verifier (8 * 2^-15 * max|golden|, 1e-4 free) took it."""
    from deepgemm_ascend_amd.harness import files
    monkeypatch.chdir(tmp_path)
    m, n, k = 256, 384, 1024
    (a, sfa), (b, sfb), golden = files.gen_golden_data(m, n, k, mode="fp8", seed=2)
    assert _run(tmp_path, [0, m, n, k, 1, 1, 3, 8, 20, 10]).returncode == 0
    assert files.verify_result("output/output.bin", "output/golden.bin", mode="fp8", policy="fast")
    raw = np.fromfile("output/output.bin", dtype=np.uint16)
    s = files.abs_term_sum_fp8(a, sfa, b, sfb).reshape(-1)
    i = int(np.argmax(s))
    val = (raw.astype(np.uint32) << 16).view(np.float32)
    bumped = np.float32(val[i] + 3 * abs(val[i]) * 2.0 ** -7 + 3 * 2.0 ** -15 * s[i])
    bad = raw.copy(); bad[i] = np.uint16(np.array([bumped], np.float32).view(np.uint32)[0] >> 16)
    bad.tofile("output/output.bin")
    assert not files.verify_result("output/output.bin", "output/golden.bin", mode="fp8", policy="fast")
    assert _run(tmp_path, [0, m, n, k, 1, 1, 3, 8, 20, 10], DGA_BF16_EXACT="1").returncode == 0
    assert files.verify_result("output/output.bin", "output/golden.bin", mode="fp8", policy="bf16_exact")


def test_bbit_verifier_takes_the_exact_policies_at_a_long_k(dga, tmp_path, monkeypatch):
    """K = 8192: the golden file is np.matmul(f32, f32) in the BLAS's summation order, 2e-5..6e-5 of whose elements are more than
    2 ULP from the oracle-order result (profiles/r03_golden_order_noise.txt).  The verifier's bar is floored at that noise
    (tolerance.check golden_order="any"), so the bit-exact strict kernel and the bf16-exact one pass it -- they would not
    pass their oracle-order bars (0 / 1e-5 of the elements) against this file."""
    from deepgemm_ascend_amd.harness import files, tolerance
    monkeypatch.chdir(tmp_path)
    m, n, k = 384, 1024, 8192
    (a, sfa), (b, sfb), golden = files.gen_golden_data(m, n, k, mode="fp8", seed=5)
    for env, policy in (({"DGA_STRICT": "1"}, "strict"), ({"DGA_BF16_EXACT": "1"}, "bf16_exact")):
        assert _run(tmp_path, [0, m, n, k, 1, 1, 3, 8, 20, 10], **env).returncode == 0
        assert files.verify_result("output/output.bin", "output/golden.bin", mode="fp8", policy=policy)
    out = (np.fromfile("output/output.bin", dtype=np.uint16).astype(np.uint32) << 16).view(np.float32).reshape(m, n)
    ok, rep = tolerance.check(out, tolerance.bf16_round(golden.reshape(m, n)), files.abs_term_sum_fp8(a, sfa, b, sfb), policy="bf16_exact")
    assert rep["worst_excess_over_S"] <= 2.0 ** -21                          # whatever the oracle-order bar says about this golden


def test_bbit_argument_errors(dga, tmp_path):
    assert _run(tmp_path, [0, 16, 16, 16]).returncode == 2                   # argc != 11 (benchmark_util.h:50)
    assert _run(tmp_path, [0, 16, 16, 16, 1, 1, 0, 8, 20, 10]).returncode == 2  # zero knob
    assert _run(tmp_path, [0, 16, 16, 16, 1, 1, 3, 8, 20, 10]).returncode == 3  # missing input files
