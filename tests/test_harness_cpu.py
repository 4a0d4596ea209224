"""CPU: file golden / verify tools (mirrors what /root/reference/deep_gemm_ascend/scripts/tests/test_all_scripts.py
asserts of gen_golden.py / verify.py: sizes, dtypes, golden == fp32 matmul, verifier accept / reject)."""
import numpy as np
import pytest

from deepgemm_ascend_amd.harness import files, sweep


def test_gen_fp16_reference_format(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)
    x1, x2, golden = files.gen_golden_data(32, 48, 16, mode="fp16", seed=0)
    assert (tmp_path / "input/x1_gm.bin").stat().st_size == 32 * 16 * 2
    assert (tmp_path / "input/x2_gm.bin").stat().st_size == 16 * 48 * 2
    g = np.fromfile(tmp_path / "output/golden.bin", np.float32).reshape(32, 48)
    assert np.array_equal(g, np.matmul(x1.astype(np.float32), x2.astype(np.float32)))  # gen_golden.py:14-15
    assert x1.min() >= 1 and x1.max() <= 10


def test_gen_fp8_files_and_golden_matches_oracle(tmp_path, monkeypatch, oracle):
    monkeypatch.chdir(tmp_path)
    (a, sfa), (b, sfb), golden = files.gen_golden_data(40, 130, 300, mode="fp8", seed=2)
    assert (tmp_path / "input/x1_gm.bin").stat().st_size == 40 * 300
    assert (tmp_path / "input/x2_gm.bin").stat().st_size == 130 * 300
    assert (tmp_path / "input/sfa.bin").stat().st_size == 40 * 3 * 4
    assert (tmp_path / "input/sfb.bin").stat().st_size == 2 * 3 * 4
    want = oracle.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb)
    assert oracle.bf16_ulp_diff(oracle.f32_to_bf16_bits(golden), want).max() <= 1
    assert np.array_equal(files.e4m3fn_table(), oracle.e4m3fn_table(), equal_nan=True)
    qa, qsf = oracle.quant_1x128(np.random.default_rng(0).standard_normal((8, 256)).astype(np.float32))
    fa, fsf = files.quant_blocks(np.random.default_rng(0).standard_normal((8, 256)).astype(np.float32), 1)
    assert np.array_equal(qa, fa) and np.array_equal(qsf, fsf)


def test_verify_semantics(tmp_path):
    g = np.linspace(1, 2, 4096, dtype=np.float32)
    g.tofile(tmp_path / "g.bin")
    (g + 1e-7).astype(np.float32).tofile(tmp_path / "ok.bin")
    (g * 2).tofile(tmp_path / "bad.bin")
    g[:100].tofile(tmp_path / "short.bin")
    np.zeros(0, np.float32).tofile(tmp_path / "empty.bin")
    assert files.verify_result(tmp_path / "ok.bin", tmp_path / "g.bin", mode="fp16")
    assert not files.verify_result(tmp_path / "bad.bin", tmp_path / "g.bin", mode="fp16")
    assert not files.verify_result(tmp_path / "short.bin", tmp_path / "g.bin", mode="fp16")   # reference crashes here
    assert files.verify_result(tmp_path / "empty.bin", tmp_path / "empty.bin", mode="fp16")     # ... and here


def test_sweep_candidates_and_shapes():
    assert len(sweep.SHAPE_GROUP) == 18 and sweep.SHAPE_GROUP[0] == [4096, 4096, 4096]   # benchmark.py:24-44
    c = sweep.candidates(4096, 4096, 4096)
    assert {"m1": 256, "n1": 256, "raster": 8, "stages": 2, "splitk": 1, "policy": 0} in c
    assert any(x["splitk"] > 1 for x in sweep.candidates(8, 7168, 18432))
    assert all(x["m1"] <= 16 for x in sweep.candidates(8, 7168, 18432))
