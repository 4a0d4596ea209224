"""The golden that the reference's OWN generator wrote (tests/golden/ref_gen_golden_96x160x320.npz: the images of
input/x1_gm.bin, input/x2_gm.bin and output/golden.bin produced by importing
/root/reference/deep_gemm_ascend/scripts/gen_golden.py in tests/golden/make_golden.py), against every path of the build that
takes the reference's own dtypes -- with the reference's own verifier thresholds:

  fp16 sweep gate                  rtol 1.5e-6, atol 1e-9, at most 1e-4 of the elements off -- the threshold the reference applies
                                   to run_mmad_bench's output          (framework/benchmark/benchmark.py:20-22,307-308,384-398)
  file verifier                    rtol 1e-6 (scripts/verify.py:10-35): met by the CPU oracle (k-ascending fp32 chain); the
                                   fp16 matrix instruction sums a 32-wide k step in its own order and leaves 5e-4 of these
                                   elements between 1e-6 and 1.5e-6 (measured, gpurun_out/r03/gputest_a.log) -- so the GPU
                                   paths are held to the sweep gate, the reference's own bar for the kernel output
  16-bit outputs (the aclnn op)    one unit in the last place of the output dtype on top (the reference has no fixture
                                   for its fp16-out operator; its bf16-input tolerance is 2e-4, framework/tests/test.py:19-21)
"""
import os
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
FIXTURE = ROOT / "tests" / "golden" / "ref_gen_golden_96x160x320.npz"
RTOL_FILE, RTOL_SWEEP = 1e-6, 1.5e-6


def _fx():
    d = np.load(FIXTURE)
    return d["x1_gm"], d["x2_gm"], d["golden"]


def test_fixture_is_what_the_reference_generator_writes():
    """Shape, dtypes and the formula of gen_golden.py:10-23; the restated generator of the product harness
    (harness/files.py) and the oracle's golden formula reproduce the reference's golden bit for bit."""
    x1, x2, golden = _fx()
    assert x1.dtype == np.float16 and x2.dtype == np.float16 and golden.dtype == np.float32
    assert x1.shape == (96, 320) and x2.shape == (320, 160) and golden.shape == (96, 160)
    assert x1.min() >= 1 and x1.max() <= 10
    assert np.array_equal(np.matmul(x1.astype(np.float32), x2.astype(np.float32)).astype(np.float32), golden)


def test_oracle_mmad_golden_meets_the_reference_verifier(oracle):
    """The oracle's fp32-accumulate CPU path (k ascending) against the reference's BLAS-ordered golden: the file verifier's
    own threshold."""
    x1, x2, golden = _fx()
    got = oracle.matmul_f32_nn(x1.astype(np.float32), x2.astype(np.float32))   # dga_oracle_matmul_f32_nn: fp32 chain, k ascending
    assert got.dtype == np.float32 and got.shape == golden.shape
    # the same chain restated in numpy (generate_code.hpp:216,320-335): the C oracle is that chain, bit for bit
    acc = np.zeros(golden.shape, np.float32)
    a32, b32 = x1.astype(np.float32), x2.astype(np.float32)
    for kk in range(x1.shape[1]):
        acc += a32[:, kk:kk + 1] * b32[kk:kk + 1, :]
    assert np.array_equal(got, acc)
    ok, ratio = oracle.verify_isclose(got, golden, rtol=RTOL_FILE)
    assert ok, ratio


@pytest.mark.gpu
def test_run_mmad_bench_on_the_reference_golden(dga, oracle):
    """run_mmad_bench (python_api.cpp:23): fp16 x [M,K], y [K,N] -> fp32 z, verified as the reference's sweep verifies it."""
    import torch
    x1, x2, golden = _fx()
    x = torch.from_numpy(x1).cuda(); y = torch.from_numpy(x2).cuda()
    z = torch.zeros(golden.shape, dtype=torch.float32, device="cuda")
    params = torch.zeros(28, dtype=torch.int32, device="cuda")
    params[:6] = torch.tensor([1, 1, 3, 8, 20, 10], dtype=torch.int32)
    dga.run_mmad_bench(x, y, z, params)
    ok, ratio = oracle.verify_isclose(z.cpu().numpy(), golden, rtol=RTOL_SWEEP)
    assert ok, ratio
    # the batched entry point on the same data (run_mmad_rtc, python_api.cpp:18)
    zb = torch.zeros((1,) + golden.shape, dtype=torch.float32, device="cuda")
    dga.run_mmad_rtc(x[None], y[None], zb)
    ok, ratio = oracle.verify_isclose(zb[0].cpu().numpy(), golden, rtol=RTOL_SWEEP)
    assert ok, ratio


@pytest.mark.gpu
def test_catlass_dynamic_matmul_on_the_reference_golden(dga, oracle):
    """The aclnn operator's own contract (fp16 in AND out, NT): mat2 = the transposed view of the contiguous [N,K] copy of
    x2.  An fp16 output carries 11 significant bits: golden rounded to fp16 is the best any kernel can do, and one unit in
    the last place (2^-10 relative) on top of the reference's 2e-4 is the bar."""
    import torch
    x1, x2, golden = _fx()
    self_ = torch.from_numpy(x1).cuda()
    mat2 = torch.from_numpy(np.ascontiguousarray(x2.T)).cuda().t()
    out = torch.zeros(golden.shape, dtype=torch.float16, device="cuda")
    dga.catlass_dynamic_matmul(self_, mat2, out, sync=True)
    ok, ratio = oracle.verify_isclose(out.float().cpu().numpy(), golden, rtol=2e-4 + 2.0 ** -10)
    assert ok, ratio
    want = golden.astype(np.float16)
    ulps = np.abs(out.cpu().numpy().view(np.int16).astype(np.int32) - want.view(np.int16).astype(np.int32))
    assert int(ulps.max()) <= 1, int(ulps.max())


@pytest.mark.gpu
def test_bbit_fp16_mode_on_the_reference_golden(dga, tmp_path):
    """dga_kernels_bbit in the reference's own file format: the three files as gen_golden.py wrote them, output.bin checked
    by the file verifier's logic at the sweep gate's threshold (see the module docstring)."""
    from deepgemm_ascend_amd.harness import files
    x1, x2, golden = _fx()
    (tmp_path / "input").mkdir(); (tmp_path / "output").mkdir()
    x1.tofile(tmp_path / "input" / "x1_gm.bin"); x2.tofile(tmp_path / "input" / "x2_gm.bin")
    golden.tofile(tmp_path / "output" / "golden.bin")
    env = dict(os.environ, DGA_BBIT_ITERS="3", DGA_BBIT_WARMUP="1")
    m, k = x1.shape
    n = x2.shape[1]
    r = subprocess.run([str(ROOT / "deepgemm_ascend_amd" / "dga_kernels_bbit"), "0", str(m), str(n), str(k), "1", "1", "3", "8", "20", "10"],
                       cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    cwd = os.getcwd()
    os.chdir(tmp_path)
    try:
        assert files.verify_result("output/output.bin", "output/golden.bin", mode="fp16", rtol=RTOL_SWEEP)
    finally:
        os.chdir(cwd)
