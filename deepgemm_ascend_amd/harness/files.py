"""gen_golden / verify for the bbit harness -- same file names, dtypes and tolerance logic as the reference's
scripts (/root/reference/deep_gemm_ascend/scripts/gen_golden.py:10-23, verify.py:14-35), extended with the
fp8 block-scaled mode.

  python -m deepgemm_ascend_amd.harness.files gen M N K [--mode fp8|fp16] [--seed S]
  python -m deepgemm_ascend_amd.harness.files verify output/output.bin output/golden.bin [--mode fp8|fp16]

fp16 mode is the reference's own format: x1 fp16 [M,K], x2 fp16 [K,N], golden = np.matmul(f32, f32) as fp32.
fp8 mode: x1 = A e4m3fn [M,K], x2 = B e4m3fn [N,K], sfa.bin / sfb.bin fp32; golden.bin = fp32 [M,N] from the
same golden formula applied per 128-wide k block (this is the harness's verifier, like the reference's
gen_golden.py -- not an operator path).
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

# verify.py:10-12 (file verifier), test.py:19-21 (bf16)
RTOL_FP32_FILE = 1e-6
ATOL = 1e-9
ERROR_TOL = 1e-4


def e4m3fn_table() -> np.ndarray:
    v = np.arange(256)
    s = np.where(v & 0x80, -1.0, 1.0)
    e = (v >> 3) & 0xF
    m = v & 7
    val = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e - 7.0))
    val = np.where((e == 15) & (m == 7), np.nan, val)
    return (s * val).astype(np.float32)


def quantize_e4m3fn(x: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even onto the e4m3fn grid, saturating at +-448 (|x| <= 448 expected)."""
    tab = e4m3fn_table()[:127].astype(np.float64)          # non-negative finite values, ascending
    ax = np.minimum(np.abs(x.astype(np.float64)), 448.0)
    hi = np.searchsorted(tab, ax, side="left").clip(1, 126)
    lo = hi - 1
    dl, dh = ax - tab[lo], tab[hi] - ax
    pick_hi = (dh < dl) | ((dh == dl) & (hi % 2 == 0))
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return (code | np.where(np.signbit(x), 0x80, 0).astype(np.uint8)).astype(np.uint8)


def quant_blocks(x: np.ndarray, row_block: int):
    """amax scaling per (row_block x 128) block -> (codes u8, scales f32)."""
    r, k = x.shape
    rb, kb = -(-r // row_block), -(-k // 128)
    q = np.zeros((r, k), np.uint8)
    sf = np.ones((rb, kb), np.float32)
    for i in range(rb):
        for j in range(kb):
            blk = x[i * row_block:(i + 1) * row_block, j * 128:(j + 1) * 128]
            amax = float(np.abs(blk).max()) if blk.size else 0.0
            s = np.float32(amax / 448.0) if amax > 0 else np.float32(1.0)
            sf[i, j] = s
            q[i * row_block:(i + 1) * row_block, j * 128:(j + 1) * 128] = quantize_e4m3fn(blk / s)
    return q, sf


def golden_fp8(a, sfa, b, sfb) -> np.ndarray:
    tab = e4m3fn_table()
    m, k = a.shape
    n = b.shape[0]
    acc = np.zeros((m, n), np.float32)
    col_blk = np.arange(n) // 128
    for kb in range(-(-k // 128)):
        sl = slice(kb * 128, min(k, kb * 128 + 128))
        part = np.matmul(tab[a[:, sl]].astype(np.float32), tab[b[:, sl]].astype(np.float32).T)
        acc += part * (sfa[:, kb][:, None] * sfb[col_blk, kb][None, :])
    return acc


def abs_term_sum_fp8(a, sfa, b, sfb) -> np.ndarray:
    """S[m, n] = sum_kb |sfa * sfb| * sum_k |a| |b| (fp64): the magnitude the parity bar's eps term scales with (tolerance.py)."""
    tab = np.abs(np.nan_to_num(e4m3fn_table().astype(np.float64), nan=0.0))
    m, k = a.shape
    n = b.shape[0]
    out = np.zeros((m, n))
    col_blk = np.arange(n) // 128
    for kb in range(-(-k // 128)):
        sl = slice(kb * 128, min(k, kb * 128 + 128))
        out += np.abs(sfa[:, kb].astype(np.float64)[:, None] * sfb[col_blk, kb].astype(np.float64)[None, :]) * (tab[a[:, sl]] @ tab[b[:, sl]].T)
    return out


def _fp8_inputs_of(golden_size: int, input_dir: str):
    """The fp8 harness files beside a golden of M*N elements: shapes recovered from the file sizes (x1 = M*K bytes,
    x2 = N*K bytes, sfa = M*KB floats, sfb = NB*KB floats)."""
    try:
        a = np.fromfile(os.path.join(input_dir, "x1_gm.bin"), dtype=np.uint8)
        b = np.fromfile(os.path.join(input_dir, "x2_gm.bin"), dtype=np.uint8)
        sfa = np.fromfile(os.path.join(input_dir, "sfa.bin"), dtype=np.float32)
        sfb = np.fromfile(os.path.join(input_dir, "sfb.bin"), dtype=np.float32)
    except OSError:
        return None
    if a.size == 0 or b.size == 0 or golden_size == 0:
        return None
    m = int(round((a.size * golden_size / b.size) ** 0.5))
    if m <= 0 or golden_size % m or a.size % m:
        return None
    n, k = golden_size // m, a.size // m
    kb, nb = -(-k // 128), -(-n // 128)
    if n * k != b.size or sfa.size != m * kb or sfb.size != nb * kb:
        return None
    return a.reshape(m, k), sfa.reshape(m, kb), b.reshape(n, k), sfb.reshape(nb, kb)


def gen_golden_data(M: int, N: int, K: int, mode: str = "fp8", seed=None, data: str = "uniform"):
    rng = np.random.default_rng(seed)
    os.makedirs("input", exist_ok=True)
    os.makedirs("output", exist_ok=True)
    if mode == "fp16":
        if data == "heavy_tail":   # benchmark.py:350-352
            gen = lambda shape: np.clip(rng.lognormal(1.0, 1.2, size=shape), 1, 10).astype(np.float16)
        else:                      # gen_golden.py:11-12
            gen = lambda shape: rng.uniform(1, 10, shape).astype(np.float16)
        x1, x2 = gen([M, K]), gen([K, N])
        golden = np.matmul(x1.astype(np.float32), x2.astype(np.float32)).astype(np.float32)
        x1.tofile("input/x1_gm.bin"); x2.tofile("input/x2_gm.bin"); golden.tofile("output/golden.bin")
        for f in ("input/sfa.bin", "input/sfb.bin"):
            if os.path.exists(f):
                os.remove(f)
        return x1, x2, golden
    xa = rng.standard_normal((M, K)).astype(np.float32)
    xb = rng.standard_normal((N, K)).astype(np.float32)
    a, sfa = quant_blocks(xa, 1)
    b, sfb = quant_blocks(xb, 128)
    golden = golden_fp8(a, sfa, b, sfb)
    a.tofile("input/x1_gm.bin"); b.tofile("input/x2_gm.bin")
    sfa.tofile("input/sfa.bin"); sfb.tofile("input/sfb.bin")
    golden.tofile("output/golden.bin")
    return (a, sfa), (b, sfb), golden


def verify_result(output_path: str, golden_path: str, mode: str = "fp8", rtol=None, policy: str = "fast",
                  input_dir: str = "input") -> bool:
    """verify.py:14-35.  fp8 mode: the product's ONE parity bar (harness/tolerance.py) -- every element within
    2 ulp_bf16 + eps(policy) * S of the bf16-rounded golden, S computed from the harness's input files."""
    golden = np.fromfile(golden_path, dtype=np.float32).reshape(-1)
    if mode == "fp8":
        from . import tolerance
        raw = np.fromfile(output_path, dtype=np.uint16)
        output = (raw.astype(np.uint32) << 16).view(np.float32).reshape(-1)
        if output.size != golden.size:
            print(f"size mismatch output={output.size}, golden={golden.size}")
            return False
        if golden.size == 0:
            print("error ratio: 0.000000 (empty)")
            return True
        ins = _fp8_inputs_of(golden.size, input_dir)
        if ins is None:
            print(f"[ERROR] {input_dir}/x1_gm.bin, x2_gm.bin, sfa.bin, sfb.bin do not match a golden of {golden.size} elements")
            return False
        a, sfa, b, sfb = ins
        s = abs_term_sum_fp8(a, sfa, b, sfb).reshape(-1)
        ok, rep = tolerance.check(output, tolerance.bf16_round(golden), s, policy=policy, short_k=a.shape[1] < 128, golden_order="any")
        if not ok:
            want = tolerance.bf16_round(golden)
            bad = np.where(np.abs(output - want) > 2 * np.abs(want) * 2.0 ** -7 + rep["eps"] * s)[0]
            for idx in bad[:100]:
                g, o = golden[idx], output[idx]
                print(f"index={idx:06d}  expect={g:-.9f}  actual={o:-.9f}  rdiff={abs(o - g) / abs(g) if g != 0 else abs(o):-.6f}")
        print(f"error ratio: {rep['frac_gt_2ulp']:.6f}  (beyond 2 ulp; allowed {rep['frac_allowed']:g}); worst excess "
              f"{rep['worst_excess_over_S']:.3e} * S (allowed {rep['eps']:.3e}); max ulp {rep['max_ulp']:.1f}")
        return ok
    else:
        output = np.fromfile(output_path, dtype=np.float32).reshape(-1)
        rtol = RTOL_FP32_FILE if rtol is None else rtol
        atol = ATOL
    if output.size != golden.size:        # the reference broadcasts and crashes here under numpy 2
        print(f"size mismatch output={output.size}, golden={golden.size}")
        return False
    if golden.size == 0:                  # ... and divides by zero here: defined as a pass
        print("error ratio: 0.000000 (empty)")
        return True
    close = np.isclose(output, golden, rtol=rtol, atol=atol, equal_nan=True)
    bad = np.where(~close)[0]
    for i, idx in enumerate(bad[:100]):
        g, o = golden[idx], output[idx]
        print(f"index={idx:06d}  expect={g:-.9f}  actual={o:-.9f}  rdiff={abs(o - g) / abs(g) if g != 0 else abs(o):-.6f}")
    ratio = bad.size / golden.size
    print(f"error ratio: {ratio:.6f}  (tolerance: {ERROR_TOL})")
    return ratio <= ERROR_TOL


def main(argv=None):
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    g = sub.add_parser("gen"); g.add_argument("M", type=int); g.add_argument("N", type=int); g.add_argument("K", type=int)
    g.add_argument("--mode", default="fp8", choices=["fp8", "fp16"]); g.add_argument("--seed", type=int, default=None)
    v = sub.add_parser("verify"); v.add_argument("output"); v.add_argument("golden")
    v.add_argument("--mode", default="fp8", choices=["fp8", "fp16"])
    a = ap.parse_args(argv)
    if a.cmd == "gen":
        gen_golden_data(a.M, a.N, a.K, a.mode, a.seed)
        print(f"generated M={a.M}, N={a.N}, K={a.K} ({a.mode})")
        return 0
    ok = verify_result(a.output, a.golden, a.mode)
    print("test pass" if ok else "[ERROR] result error")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
