#!/usr/bin/env python3
"""Generates the fixtures under tests/golden/.  Run in the authoring container only
(`python tests/golden/make_golden.py`): it imports the reference's Python autotuner mirror from
/root/reference/get_best_config and runs oracle/_ref/ref_config (the reference's get_best_config.hpp
compiled where it lies).  Only DATA is written here (inputs + expected outputs); no reference source.

Fixtures (SURVEY.md 8c "Golden vectors to commit"):
  e4m3fn_table.json       (i)   256-entry decode table, from torch.float8_e4m3fn (an independent implementation)
  c1_unit_128.npz         (ii)  BASELINE config 1: 128^3, unit scales; expected = the reference golden formula
                                np.matmul(x1.astype(f32), x2.astype(f32)) (framework/tests/test.py:37) + bf16 RNE
  scaled_64x256x400.npz   (iii) non-unit scales and a K tail block; expected from the oracle (parity unpinned)
  grouped_g4_m16.npz      (iv)  G=4, M_max=16, masked_m=[0,1,7,16]; expected from the oracle (parity unpinned)
  config_vectors.json     (v)   28-int Config tuples from the reference header itself
  op_tiling_vectors.json  (vi)  (m1,n1,k1,kernelSerial,blockDim,padding) from the reference's Python mirror
                                (coreNum 20) and the C++ probes recorded in SURVEY.md 8(a7) (coreNum 24)
  ref_gen_golden_96x160x320.npz (vii) the files the reference's OWN generator writes: deep_gemm_ascend/scripts/gen_golden.py
                                is imported, numpy's global generator seeded, gen_golden_data(96, 160, 320) run in a scratch
                                directory; the bytes of input/x1_gm.bin, input/x2_gm.bin and output/golden.bin are stored.
                                `python tests/golden/make_golden.py ref_gen_golden` writes this one alone.
  ref_gen_golden_e4m3_96x160x320.npz (viii) the same generator with numpy.random.uniform returning draws from the signed e4m3fn
                                grid for the duration of the call: a reference-written golden that reaches the FP8 kernels
                                (`... make_golden.py ref_gen_golden_e4m3`).
  select_strategy_vectors.json  (ix) picks of the reference predictor's selection strategies (greedy / topk_median / topk_dbscan):
                                get_best_config.py's TilingPredictor.select_tiling_strategy imported and called on seeded lists
                                (`... make_golden.py select_strategy`).
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import oracle as O  # noqa: E402

REF = Path("/root/reference")


def table():
    import torch
    t = torch.arange(256, dtype=torch.uint8).view(torch.float8_e4m3fn).float().numpy()
    (HERE / "e4m3fn_table.json").write_text(json.dumps(
        {"source": "torch.float8_e4m3fn", "f32_bits_hex": [f"{int(v):08x}" for v in t.view(np.uint32)]}))


def c1():
    a, sfa, b, sfb = O.make_inputs(128, 128, 128, seed=0, unit_scales=True)
    tab = O.np_e4m3fn_table()
    x1 = tab[a].astype(np.float32); x2 = tab[b].astype(np.float32).T          # [M,K], [K,N]
    golden = np.matmul(x1.astype(np.float32), x2.astype(np.float32)).astype(np.float32)  # test.py:37
    np.savez_compressed(HERE / "c1_unit_128.npz", a=a, b=b, sfa=sfa, sfb=sfb, golden_f32=golden,
                        expected_bf16=O.f32_to_bf16_bits(golden))


def scaled():
    a, sfa, b, sfb = O.make_inputs(64, 256, 400, seed=3)
    out, acc = O.gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, want_f32=True)
    np.savez_compressed(HERE / "scaled_64x256x400.npz", a=a, b=b, sfa=sfa, sfb=sfb, expected_bf16=out, acc_f32=acc)


def grouped():
    g, mmax, n, k = 4, 16, 128, 256
    parts = [O.make_inputs(mmax, n, k, seed=40 + i) for i in range(g)]
    a, sfa, b, sfb = (np.stack([p[j] for p in parts]) for j in range(4))
    masked = np.array([0, 1, 7, 16], np.int32)
    init = np.full((g, mmax, n), 0x7FC1, np.uint16)
    out = O.m_grouped_gemm_fp8_fp8_bf16_nt_masked(a, sfa, b, sfb, init, masked)
    np.savez_compressed(HERE / "grouped_g4_m16.npz", a=a, b=b, sfa=sfa, sfb=sfb, masked_m=masked, init=init,
                        expected_bf16=out)


def configs():
    best = [(1, 4096, 4096, 4096), (1, 4096, 2048, 7168), (1, 1, 512, 128), (2, 100, 200, 300), (1, 128, 128, 128),
            (3, 17, 33, 65), (1, 1279, 5003, 7681), (1, 16, 16, 16), (1, 48, 128, 320)]
    bench = [(96, 1536, 5952, 1, 1, 3, 8, 20, 10), (4096, 4096, 4096, 4, 6, 8, 8, 16, 8), (64, 4096, 7168, 1, 24, 4, 16, 12, 4),
             (1279, 5003, 7681, 3, 8, 5, 7, 9, 3), (8, 7168, 18432, 1, 20, 1, 32, 15, 5), (33, 65, 129, 2, 2, 1, 1, 1, 1),
             (1024, 4096, 7168, 2, 12, 16, 8, 10, 10)]
    out = {"order": "struct Config declaration order, get_best_config.hpp:12-31",
           "best": [{"args": list(a), "config": O.ref_config("best", *a)} for a in best],
           "bench": [{"args": list(a), "config": O.ref_config("bench", *a)} for a in bench]}
    (HERE / "config_vectors.json").write_text(json.dumps(out, indent=0))


def op_tiling():
    sys.path.insert(0, str(REF / "get_best_config"))
    import tiling_calculator as tc  # the reference's own module (imported, never copied)
    shapes = [(128, 128, 128), (4096, 4096, 4096), (4096, 2048, 7168), (128, 2048, 7168), (8, 7168, 18432),
              (8, 18432, 7168), (64, 4096, 7168), (64, 7168, 18432), (64, 24576, 1536), (64, 32768, 512),
              (128, 4096, 7168), (1024, 4096, 7168), (1024, 18432, 7168), (2048, 4096, 7168), (1279, 5003, 7681),
              (3511, 6151, 8191), (5119, 6997, 9901), (16, 16, 16), (256, 256, 256), (512, 512, 512), (1, 512, 128),
              (300, 200, 100), (2000, 100, 4000), (100, 2000, 4000), (4096, 4096, 128), (768, 768, 4096)]
    rows = []
    for core in (20, 24):
        pf = tc.PlatformInfo(); pf.coreNum = core
        calc = tc.MatmulTilingCalculator(pf)
        for (m, n, k) in shapes:
            r = calc.calculate(m, n, k, tc.LayoutTag.TagRowMajor, tc.LayoutTag.TagColumnMajor)
            dt = list(calc.calculate_tiling(m, n, k, tc.LayoutTag.TagRowMajor, tc.LayoutTag.TagColumnMajor))
            # The Python mirror has a branch the C++ op_tiling lacks (tiling_calculator.py "m1t_alt": it swaps to
            # 256x128 when the *alternative* stream-K tile would qualify, then falls through to Common).  Rows where it
            # fired are flagged; the C++ (select_kernel.cpp:333-369), which is what we restate, keeps `do_tiling`.
            alt = (list(r["tiling"]) != dt) and int(r["kernelSerial"]) in (0, 2)
            rows.append({"coreNum": core, "shape": [m, n, k], "tiling": list(r["tiling"]), "do_tiling": dt,
                         "python_only_alt_branch": bool(alt),
                         "kernelSerial": int(r["kernelSerial"]), "blockDim": int(r["blockDim"]),
                         "padding": [int(r["paddingTagA"]), int(r["paddingTagB"]), int(r["paddingTagC"])],
                         "operator_type": r["operator_type"]})
    survey = [  # C++ op_tiling probes quoted in SURVEY.md 8(a7), coreNum 24
        {"shape": [128, 128, 128], "tiling": [64, 64, 1024], "kernelSerial": 1, "blockDim": 4},
        {"shape": [4096, 4096, 4096], "tiling": [128, 256, 256], "kernelSerial": 0, "blockDim": 24},
        {"shape": [4096, 2048, 7168], "tiling": [256, 128, 256], "kernelSerial": 0, "blockDim": 24},
        {"shape": [128, 2048, 7168], "tiling": [128, 96, 512], "kernelSerial": 0, "blockDim": 22}]
    (HERE / "op_tiling_vectors.json").write_text(json.dumps(
        {"layout": "NT (A row-major, B column-major)", "python_mirror": rows, "survey_cpp_probes": survey}, indent=0))


def ref_gen_golden(m=96, n=160, k=320, seed=20251004):
    """The reference's own golden generator, imported and run (data only is kept: three file images)."""
    import contextlib
    import io
    import os
    import tempfile
    sys.path.insert(0, str(REF / "deep_gemm_ascend" / "scripts"))
    import gen_golden as ref   # the reference's module (imported, never copied)
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        try:
            np.random.seed(seed)           # gen_golden.py:11-12 draws from numpy's global generator
            with contextlib.redirect_stdout(io.StringIO()):
                ref.gen_golden_data(m, n, k)
            x1 = np.fromfile("input/x1_gm.bin", dtype=np.float16).reshape(m, k)
            x2 = np.fromfile("input/x2_gm.bin", dtype=np.float16).reshape(k, n)
            golden = np.fromfile("output/golden.bin", dtype=np.float32).reshape(m, n)
        finally:
            os.chdir(cwd)
    np.savez_compressed(HERE / f"ref_gen_golden_{m}x{n}x{k}.npz", x1_gm=x1, x2_gm=x2, golden=golden,
                        meta=np.array([m, n, k, seed], np.int64),
                        source=np.array("deep_gemm_ascend/scripts/gen_golden.py:10-23 gen_golden_data, np.random.seed(seed), numpy " + np.__version__))


def ref_gen_golden_e4m3(m=96, n=160, k=320, seed=20251005):
    """The reference's own golden generator again, this time writing a golden the FP8 kernels can consume: for the duration of
    gen_golden_data(m, n, k) numpy's `random.uniform` (what gen_golden.py:11-12 draws its operands from) returns draws from the
    signed e4m3fn grid -- the decode of random bytes, NaN codes excluded -- so x1_gm.bin / x2_gm.bin hold values every one of
    which is an e4m3 number (exact in the fp16 the generator stores), and golden.bin is the reference's own
    np.matmul(f32, f32) of them.  Data only is kept: the three file images.  K = 320: two full 128-wide scale blocks and a tail."""
    import contextlib
    import io
    import os
    import tempfile
    sys.path.insert(0, str(REF / "deep_gemm_ascend" / "scripts"))
    import gen_golden as ref   # the reference's module (imported, never copied)
    tab = O.np_e4m3fn_table()
    codes = np.array([c for c in range(256) if (c & 0x7F) != 0x7F], np.uint8)
    rng = np.random.default_rng(seed)
    real_uniform = np.random.uniform

    def grid_uniform(low, high, size):   # same call shape as gen_golden.py:11-12: uniform(1, 10, [rows, cols])
        return tab[codes[rng.integers(0, codes.size, size=tuple(size))]].astype(np.float64)

    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as d:
        os.chdir(d)
        np.random.uniform = grid_uniform
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                ref.gen_golden_data(m, n, k)
            x1 = np.fromfile("input/x1_gm.bin", dtype=np.float16).reshape(m, k)
            x2 = np.fromfile("input/x2_gm.bin", dtype=np.float16).reshape(k, n)
            golden = np.fromfile("output/golden.bin", dtype=np.float32).reshape(m, n)
        finally:
            np.random.uniform = real_uniform
            os.chdir(cwd)
    np.savez_compressed(HERE / f"ref_gen_golden_e4m3_{m}x{n}x{k}.npz", x1_gm=x1, x2_gm=x2, golden=golden,
                        meta=np.array([m, n, k, seed], np.int64),
                        source=np.array("deep_gemm_ascend/scripts/gen_golden.py:10-23 gen_golden_data with numpy.random.uniform returning "
                                        "draws from the signed e4m3fn grid (default_rng(seed)), numpy " + np.__version__))


def select_strategy():
    """Golden picks of the reference predictor's selection strategies: TilingPredictor.select_tiling_strategy
    (get_best_config/get_best_config.py:431-525) imported and called on seeded candidate lists.  greedy / topk_median: the picked
    index.  topk_dbscan: the winning CLUSTER (the reference returns random.Random(random_state).choice of it; the module's `random`
    is replaced for the call by a recorder that captures the list it is asked to choose from) -- or the fallback index when no
    cluster forms."""
    sys.path.insert(0, str(REF / "get_best_config"))
    import get_best_config as gbc   # the reference's module (imported, never copied)
    rng = np.random.default_rng(20251006)
    captured = []

    class Recorder:
        def __init__(self, seed=None):
            pass

        def choice(self, seq):
            captured.append(list(seq))
            return seq[0]

    class FakeRandom:
        Random = Recorder

    cases = []
    tile_values = [16, 32, 48, 64, 96, 128, 192, 256]
    for n in (1, 2, 3, 8, 15, 40, 120):
        for rep in range(4):
            # a few performance plateaus (tiles that behave alike) plus scattered outliers, no two times equal
            centres = rng.uniform(20, 400, size=max(1, n // 6 + 1))
            preds = np.array([c * rng.uniform(0.97, 1.03) for c in rng.choice(centres, size=n)], np.float32)
            preds += np.arange(n, dtype=np.float32) * 1e-3
            tiles = [[int(rng.choice(tile_values)), int(rng.choice(tile_values)), int(rng.choice([64, 128, 256, 512, 1024]))] for _ in range(n)]
            params = [{"mTile": t[0], "nTile": t[1], "kTile": t[2]} for t in tiles]
            for method, topk, eps, ms in (("greedy", 10, 0.8, 2), ("topk_median", 1, 0.8, 2), ("topk_median", 5, 0.8, 2),
                                          ("topk_median", 10, 0.8, 2), ("topk_median", 20, 0.8, 2), ("topk_dbscan", 5, 0.8, 2),
                                          ("topk_dbscan", 10, 0.8, 2), ("topk_dbscan", 20, 0.5, 2), ("topk_dbscan", 20, 1.5, 3),
                                          ("topk_dbscan", 60, 0.8, 2), ("topk_dbscan", 60, 1.2, 4)):
                del captured[:]
                real = gbc.random
                gbc.random = FakeRandom
                try:
                    got, t = gbc.TilingPredictor.select_tiling_strategy(None, params, preds, method=method, topk=topk, random_state=7,
                                                                        dbscan_eps=eps, dbscan_min_samples=ms)
                finally:
                    gbc.random = real
                idx = params.index(got) if captured == [] else None
                # (params may hold equal dicts: identify the pick by its predicted time, which is unique)
                idx = int(np.nonzero(preds == np.float32(t))[0][0]) if captured == [] else None
                cases.append({"preds": [float(x) for x in preds], "tiles": tiles, "method": method, "topk": topk, "eps": eps,
                              "min_samples": ms, "index": idx, "cluster": [int(i) for i in captured[0]] if captured else None})
    (HERE / "select_strategy_vectors.json").write_text(json.dumps(
        {"source": "get_best_config/get_best_config.py:431-525 TilingPredictor.select_tiling_strategy, numpy " + np.__version__,
         "cases": cases}))


if __name__ == "__main__":
    if sys.argv[1:] == ["select_strategy"]:
        select_strategy()
    elif sys.argv[1:] == ["ref_gen_golden"]:
        ref_gen_golden()
    elif sys.argv[1:] == ["ref_gen_golden_e4m3"]:
        ref_gen_golden_e4m3()
    else:
        O.build()
        table(); c1(); scaled(); grouped(); configs(); op_tiling(); ref_gen_golden(); ref_gen_golden_e4m3(); select_strategy()
    print("fixtures written to", HERE)
