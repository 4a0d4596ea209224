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


def test_sweep_constraint_checkers():
    """harness/sweep.check_candidate: the counterpart of the reference's per-variant constraint checkers
    (get_best_config/catlass_parameter.py:308-368; test_catlass_parameter.py:142-180 pins e.g. smallmatmul k > k1 -> False)."""
    from deepgemm_ascend_amd.harness import sweep
    c = lambda m1, n1, wm, wn, st, pol=0, sk=1: {"m1": m1, "n1": n1, "wavesM": wm, "wavesN": wn, "stages": st, "policy": pol, "splitk": sk}
    masked = {"m": 128, "n": 2048, "k": 7168, "groups": 256, "layout": "masked", "rows_per_group": 128}
    assert sweep.check_candidate(masked, c(128, 256, 2, 2, 3, 4)) == (True, "")
    assert not sweep.check_candidate(masked, c(64, 256, 1, 4, 3))[0]            # two tile rows would read B twice
    assert not sweep.check_candidate(masked, c(128, 256, 2, 2, 3, 0, 2))[0]     # split-K is dense only
    assert not sweep.check_candidate(masked, c(128, 256, 2, 2, 2, 4))[0]        # loader waves exist for the 3-stage build only
    assert not sweep.check_candidate(masked, c(256, 256, 4, 2, 3))[0]           # no such build (3 x 64 KB > LDS)
    small = dict(masked, m=16, rows_per_group=16)
    assert sweep.check_candidate(small, c(16, 128, 1, 4, 2))[0] and not sweep.check_candidate(small, c(32, 128, 1, 4, 2))[0]
    cont = {"m": 32 * 128, "n": 4096, "k": 7168, "groups": 32, "layout": "contiguous", "rows_per_group": 128}
    assert not sweep.check_candidate(cont, c(256, 256, 4, 2, 2, 2))[0]          # a 256-row tile would straddle two 128-row groups
    assert sweep.check_candidate(dict(cont, groups=8, rows_per_group=1024, m=8192), c(256, 256, 4, 2, 2, 2))[0]
    dense = {"m": 64, "n": 4096, "k": 7168, "groups": 1, "layout": "dense", "rows_per_group": 64}
    assert not sweep.check_candidate(dense, c(256, 256, 4, 2, 2, 2))[0]         # tile twice the problem
    assert sweep.check_candidate(dense, c(64, 128, 1, 4, 2, 0, 8))[0]           # split-K: few tiles, long K
    assert not sweep.check_candidate(dict(dense, k=512), c(64, 128, 1, 4, 2, 0, 8))[0]   # < 4 k blocks per split
    # every build of the menu fits the LDS and the accumulator budget by construction
    for (bm, bn, wm, wn, st, pols) in sweep.MENU:
        assert sweep.stage_bytes(bm, bn, wm * wn) * st <= sweep.LDS_BYTES and bm * bn // (wm * wn * 64) <= sweep.ACC_REGS
    assert len(sweep.grouped_candidates(masked)) >= 8


def test_merge_sweep_runs_and_table_builder(tmp_path):
    """scripts/merge_sweep_runs.py: the faster timing per candidate wins, a cold record replaces a warm one whatever the two
    say, the persistent forms fold into their one-tile siblings; scripts/build_tuned_table.py: first file wins, 17-column
    dense rows are padded with groups = 1, contiguous = 0."""
    import json, subprocess, sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    rec = lambda idx, t, p, neg=False: json.dumps({"idx": idx, "M": 64, "N": 512, "K": 512, "time": t, "diff": 0.0, "negative": neg, "parameters": p}) + "\n"
    base = {"m1": 64, "n1": 128, "raster": 1, "stages": 3, "splitk": 1}
    a, b, c = tmp_path / "a", tmp_path / "b", tmp_path / "c"
    for d in (a, b, c):
        d.mkdir()
    name = "shape_64_512_512_rank_0.jsonl"
    (a / name).write_text(rec(0, 20.0, dict(base, policy=4)) + rec(1, 30.0, dict(base, policy=0)) + rec(2, 25.0, dict(base, m1=256, n1=256, stages=2, policy=2)))
    (b / name).write_text(rec(0, 18.0, dict(base, policy=5)) + rec(1, 35.0, dict(base, policy=0)) + rec(2, 22.0, dict(base, m1=256, n1=256, stages=2, policy=6))
                          + rec(3, 5.0, dict(base, m1=16, policy=5)))          # no sibling record: dropped
    (c / name).write_text(rec(1, 40.0, dict(base, policy=0, cold_sets=3)))     # cold: replaces the warm 30.0
    out = tmp_path / "merged"
    subprocess.run([sys.executable, str(root / "scripts" / "merge_sweep_runs.py"), str(out), str(a), str(b), str(c)], check=True, capture_output=True)
    rows = [json.loads(l) for l in (out / name).read_text().splitlines()]
    by = {(r["parameters"]["m1"], r["parameters"]["policy"]): r for r in rows}
    assert by[(64, 4)]["time"] == 18.0 and by[(256, 2)]["time"] == 22.0          # folded persistent forms, faster wins
    assert by[(64, 0)]["time"] == 40.0 and by[(64, 0)]["parameters"].get("cold") == 1
    assert (16, 4) not in by and len(rows) == 3
    dense = tmp_path / "dense.csv"; dense2 = tmp_path / "dense2.csv"; grouped = tmp_path / "grouped.csv"
    head17 = "m,n,k,m1,n1,k1,kernelSerial,paddingTagA,paddingTagB,paddingTagC,blockDim,splitkFactor,stages,swizzleOffset,wavesM,wavesN,dispatchPolicyTag\n"
    dense.write_text(head17 + "64,512,512,64,128,128,4,0,0,0,16,4,3,1,0,0,0\n")
    dense2.write_text(head17 + "64,512,512,16,128,128,0,0,0,0,4,1,3,1,0,0,4\n" + "8,512,512,16,128,128,0,0,0,0,4,1,3,1,0,0,4\n")
    grouped.write_text(head17.strip() + ",groups,contiguous\n" + "64,512,512,64,256,128,0,0,0,0,32,1,3,1,1,4,5,16,0\n")
    table = tmp_path / "table.csv"
    subprocess.run([sys.executable, str(root / "scripts" / "build_tuned_table.py"), "--out", str(table), str(dense), str(dense2), str(grouped)],
                   check=True, capture_output=True)
    lines = table.read_text().strip().splitlines()
    assert len(lines) == 4 and all(len(l.split(",")) == 19 for l in lines)
    assert lines[1].split(",")[3:5] == ["64", "128"] and lines[1].endswith(",1,0")   # the first file's row of (64, 512, 512) won
    assert lines[3].endswith(",16,0")
