"""CPU oracle for the hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product package ``deepgemm_ascend_amd`` never does.

Two independent restatements live here:

* ``libdga_oracle.so`` (``oracle/dga_oracle.c``): scalar C, the definition of record.
* numpy forms in this file (``np_*``): the reference's golden formula
  ``np.matmul(x1.astype(f32), x2.astype(f32))``
  (/root/reference/deep_gemm_ascend/framework/tests/test.py:37,
  framework/benchmark/benchmark.py:362, scripts/gen_golden.py:14-15) applied per
  128-wide k block, used to cross-check the C code and as the BLAS-quality CPU bound.

Parity status: the fp8 decode / block-scale / bf16-rounding / masked-grouped
semantics are **parity unpinned** (the reference has no such code).
With unit scales the oracle reduces to the reference's golden formula, which is what
``tests/test_oracle.py`` pins it to.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB_PATH = _HERE / "libdga_oracle.so"
_lib = None
_lock = threading.Lock()

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_u16p = ctypes.POINTER(ctypes.c_uint16)
c_f32p = ctypes.POINTER(ctypes.c_float)
c_f64p = ctypes.POINTER(ctypes.c_double)
c_i32p = ctypes.POINTER(ctypes.c_int32)
i64 = ctypes.c_int64


def build(force: bool = False) -> Path:
    """Compile the C oracle (and oracle/_ref when /root/reference exists)."""
    src = _HERE / "dga_oracle.c"
    if force or not _LIB_PATH.exists() or _LIB_PATH.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "libdga_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    with _lock:
        if _lib is None:
            build()
            L = ctypes.CDLL(str(_LIB_PATH))
            L.dga_oracle_e4m3fn_to_f32.restype = ctypes.c_float
            L.dga_oracle_e4m3fn_to_f32.argtypes = [ctypes.c_uint8]
            L.dga_oracle_e4m3fn_table.argtypes = [c_f32p]
            L.dga_oracle_f32_to_e4m3fn.restype = ctypes.c_uint8
            L.dga_oracle_f32_to_e4m3fn.argtypes = [ctypes.c_float]
            L.dga_oracle_f32_to_bf16.restype = ctypes.c_uint16
            L.dga_oracle_f32_to_bf16.argtypes = [ctypes.c_float]
            L.dga_oracle_matmul_f32_nn.argtypes = [c_f32p, c_f32p, c_f32p, i64, i64, i64]
            L.dga_oracle_matmul_f32_nt.argtypes = [c_f32p, c_f32p, c_f32p, i64, i64, i64]
            L.dga_oracle_gemm_fp8_fp8_bf16_nt_rows.restype = ctypes.c_int
            L.dga_oracle_gemm_fp8_fp8_bf16_nt_rows.argtypes = [
                c_u8p, c_f32p, c_u8p, c_f32p, c_u16p, i64, i64, i64, i64, i64, c_f32p]
            L.dga_oracle_gemm_fp8_fp8_f64_nt.restype = ctypes.c_int
            L.dga_oracle_gemm_fp8_fp8_f64_nt.argtypes = [c_u8p, c_f32p, c_u8p, c_f32p, c_f64p, i64, i64, i64]
            L.dga_oracle_m_grouped_gemm_fp8_fp8_bf16_nt_masked.restype = ctypes.c_int
            L.dga_oracle_m_grouped_gemm_fp8_fp8_bf16_nt_masked.argtypes = [
                c_u8p, c_f32p, c_u8p, c_f32p, c_u16p, c_i32p, i64, i64, i64, i64, i64, i64]
            L.dga_oracle_quant_1x128.argtypes = [c_f32p, c_u8p, c_f32p, i64, i64]
            L.dga_oracle_quant_128x128.argtypes = [c_f32p, c_u8p, c_f32p, i64, i64]
            L.dga_oracle_quant_1x128_ex.argtypes = [c_f32p, c_u8p, c_f32p, i64, i64, ctypes.c_int]
            L.dga_oracle_quant_128x128_ex.argtypes = [c_f32p, c_u8p, c_f32p, i64, i64, ctypes.c_int]
            L.dga_oracle_verify_isclose.restype = i64
            L.dga_oracle_verify_isclose.argtypes = [c_f32p, c_f32p, i64, ctypes.c_double, ctypes.c_double, c_f64p]
            _lib = L
    return _lib


def _p(a: np.ndarray, t):
    return a.ctypes.data_as(t)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


# --------------------------------------------------------------------------- fp8 / bf16

def e4m3fn_table() -> np.ndarray:
    out = np.empty(256, np.float32)
    lib().dga_oracle_e4m3fn_table(_p(out, c_f32p))
    return out


def np_e4m3fn_table() -> np.ndarray:
    """Independent numpy statement of OCP e4m3fn (bias 7, no inf, S.1111.111 = NaN)."""
    v = np.arange(256, dtype=np.int64)
    s = np.where(v & 0x80, -1.0, 1.0)
    e = (v >> 3) & 0xF
    m = v & 0x7
    val = np.where(e == 0, m / 8.0 * 2.0 ** -6, (1 + m / 8.0) * 2.0 ** (e - 7.0))
    val = np.where((e == 0xF) & (m == 0x7), np.nan, val)
    return (s * val).astype(np.float32)


def decode_e4m3fn(q: np.ndarray) -> np.ndarray:
    return e4m3fn_table()[np.asarray(q, np.uint8)]


def bf16_bits_to_f32(h: np.ndarray) -> np.ndarray:
    return (np.asarray(h, np.uint16).astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """RNE, NaN kept NaN (same rule as dga_oracle_f32_to_bf16)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    r = np.where(nan, (u >> 16) | 0x40, r)
    return r.astype(np.uint16)


def bf16_ulp_diff(x_bits: np.ndarray, y_bits: np.ndarray) -> np.ndarray:
    """Distance in bf16 ULPs between two bf16 bit patterns (monotone integer map;
    +0/-0 coincide; NaN vs NaN = 0, NaN vs number = a huge value)."""
    def key(b):
        b = np.asarray(b, np.uint16).astype(np.int32)
        mag = b & 0x7FFF
        return np.where(b & 0x8000, -mag, mag)
    xb = np.asarray(x_bits, np.uint16)
    yb = np.asarray(y_bits, np.uint16)
    xn = (xb & 0x7FFF) > 0x7F80
    yn = (yb & 0x7FFF) > 0x7F80
    d = np.abs(key(xb) - key(yb))
    d = np.where(xn & yn, 0, d)
    d = np.where(xn ^ yn, 1 << 20, d)
    return d


# --------------------------------------------------------------------------- GEMMs

def matmul_f32_nn(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = _c(a, np.float32); b = _c(b, np.float32)
    m, k = a.shape; k2, n = b.shape
    assert k == k2
    c = np.empty((m, n), np.float32)
    lib().dga_oracle_matmul_f32_nn(_p(a, c_f32p), _p(b, c_f32p), _p(c, c_f32p), m, n, k)
    return c


def gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, threads: int = 1, want_f32: bool = False):
    """C oracle.  a [M,K] u8, sfa [M,KB] f32, b [N,K] u8, sfb [NB,KB] f32 -> bf16 bits [M,N]."""
    a = _c(a, np.uint8); b = _c(b, np.uint8); sfa = _c(sfa, np.float32); sfb = _c(sfb, np.float32)
    m, k = a.shape; n, k2 = b.shape
    assert k == k2
    kb = (k + 127) // 128; nb = (n + 127) // 128
    assert sfa.shape == (m, kb), (sfa.shape, (m, kb))
    assert sfb.shape == (nb, kb), (sfb.shape, (nb, kb))
    out = np.zeros((m, n), np.uint16)
    f32 = np.zeros((m, n), np.float32) if want_f32 else None
    fp = _p(f32, c_f32p) if want_f32 else None
    L = lib()

    def run(r0, r1):
        rc = L.dga_oracle_gemm_fp8_fp8_bf16_nt_rows(_p(a, c_u8p), _p(sfa, c_f32p), _p(b, c_u8p), _p(sfb, c_f32p),
                                                    _p(out, c_u16p), m, n, k, r0, r1, fp)
        assert rc == 0
    if threads <= 1 or m < 2:
        run(0, m)
    else:
        threads = min(threads, m)
        bounds = np.linspace(0, m, threads + 1).astype(int)
        ts = [threading.Thread(target=run, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
    return (out, f32) if want_f32 else out


def gemm_fp8_fp8_f64_nt(a, sfa, b, sfb) -> np.ndarray:
    a = _c(a, np.uint8); b = _c(b, np.uint8); sfa = _c(sfa, np.float32); sfb = _c(sfb, np.float32)
    m, k = a.shape; n, _ = b.shape
    out = np.zeros((m, n), np.float64)
    rc = lib().dga_oracle_gemm_fp8_fp8_f64_nt(_p(a, c_u8p), _p(sfa, c_f32p), _p(b, c_u8p), _p(sfb, c_f32p),
                                              _p(out, c_f64p), m, n, k)
    assert rc == 0
    return out


def np_gemm_fp8_fp8_bf16_nt(a, sfa, b, sfb, table=None) -> np.ndarray:
    """numpy restatement: the reference golden formula per 128-wide k block.
    Returns fp32 accumulator (round with f32_to_bf16_bits)."""
    tab = np_e4m3fn_table() if table is None else table
    a = np.asarray(a, np.uint8); b = np.asarray(b, np.uint8)
    m, k = a.shape; n, _ = b.shape
    kbn = (k + 127) // 128
    af = tab[a]; bf = tab[b]
    acc = np.zeros((m, n), np.float32)
    col_blk = np.arange(n) // 128
    for kb in range(kbn):
        sl = slice(kb * 128, min(k, kb * 128 + 128))
        part = np.matmul(af[:, sl].astype(np.float32), bf[:, sl].astype(np.float32).T).astype(np.float32)
        s = (np.asarray(sfa, np.float32)[:, kb][:, None] * np.asarray(sfb, np.float32)[col_blk, kb][None, :])
        acc = (acc + (part * s.astype(np.float32)).astype(np.float32)).astype(np.float32)
    return acc


def m_grouped_gemm_fp8_fp8_bf16_nt_masked(a, sfa, b, sfb, out_init, masked_m, threads: int = 1) -> np.ndarray:
    """C oracle, grouped masked-M.  out_init is copied; rows >= masked_m[g] keep its values."""
    a = _c(a, np.uint8); b = _c(b, np.uint8); sfa = _c(sfa, np.float32); sfb = _c(sfb, np.float32)
    masked_m = _c(masked_m, np.int32)
    g, mmax, k = a.shape; g2, n, k2 = b.shape
    assert g == g2 and k == k2
    out = np.array(out_init, dtype=np.uint16, copy=True, order="C")
    assert out.shape == (g, mmax, n)
    L = lib()

    def run(g0, g1):
        rc = L.dga_oracle_m_grouped_gemm_fp8_fp8_bf16_nt_masked(
            _p(a, c_u8p), _p(sfa, c_f32p), _p(b, c_u8p), _p(sfb, c_f32p), _p(out, c_u16p), _p(masked_m, c_i32p),
            g, mmax, n, k, g0, g1)
        assert rc == 0, rc
    if threads <= 1 or g < 2:
        run(0, g)
    else:
        threads = min(threads, g)
        bounds = np.linspace(0, g, threads + 1).astype(int)
        ts = [threading.Thread(target=run, args=(int(bounds[i]), int(bounds[i + 1]))) for i in range(threads)]
        [t.start() for t in ts]; [t.join() for t in ts]
    return out


def m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(a, sfa, b, sfb, out_init, m_indices, threads: int = 1) -> np.ndarray:
    """Contiguous-grouped layout: row r of a [Msum,K] is multiplied with b[m_indices[r]]; rows with a negative index
    keep out_init.  Stated row-wise on top of the dense C oracle (no alignment assumption)."""
    a = _c(a, np.uint8); sfa = _c(sfa, np.float32)
    m_indices = _c(m_indices, np.int32)
    out = np.array(out_init, dtype=np.uint16, copy=True, order="C")
    for g in range(b.shape[0]):
        rows = np.nonzero(m_indices == g)[0]
        if rows.size:
            out[rows] = gemm_fp8_fp8_bf16_nt(a[rows], sfa[rows], b[g], sfb[g], threads=threads)
    return out


# --------------------------------------------------------------------------- inputs

def quant_1x128(x: np.ndarray, ue8m0: bool = False):
    """ue8m0: scales rounded up to powers of two (2^ceil(log2(amax / 448)))."""
    x = _c(x, np.float32)
    rows, k = x.shape
    q = np.empty((rows, k), np.uint8); sf = np.empty((rows, (k + 127) // 128), np.float32)
    lib().dga_oracle_quant_1x128_ex(_p(x, c_f32p), _p(q, c_u8p), _p(sf, c_f32p), rows, k, 1 if ue8m0 else 0)
    return q, sf


def quant_128x128(x: np.ndarray, ue8m0: bool = False):
    x = _c(x, np.float32)
    rows, k = x.shape
    q = np.empty((rows, k), np.uint8); sf = np.empty(((rows + 127) // 128, (k + 127) // 128), np.float32)
    lib().dga_oracle_quant_128x128_ex(_p(x, c_f32p), _p(q, c_u8p), _p(sf, c_f32p), rows, k, 1 if ue8m0 else 0)
    return q, sf


def make_inputs(m: int, n: int, k: int, seed: int = 0, unit_scales: bool = False, ue8m0: bool = False):
    """SURVEY.md 8(d) recipe: fp32 ~ N(0,1), per-1x128 (A) / per-128x128 (B) amax scaling, cast e4m3fn.
    ue8m0: the scales rounded up to powers of two."""
    rng = np.random.default_rng(seed)
    xa = rng.standard_normal((m, k), dtype=np.float32)
    xb = rng.standard_normal((n, k), dtype=np.float32)
    if unit_scales:
        tabq = np.vectorize(lambda v: lib().dga_oracle_f32_to_e4m3fn(float(v)), otypes=[np.uint8])
        a = tabq(xa) if xa.size else xa.astype(np.uint8)
        b = tabq(xb) if xb.size else xb.astype(np.uint8)
        sfa = np.ones((m, (k + 127) // 128), np.float32)
        sfb = np.ones(((n + 127) // 128, (k + 127) // 128), np.float32)
        return a, sfa, b, sfb
    a, sfa = quant_1x128(xa, ue8m0)
    b, sfb = quant_128x128(xb, ue8m0)
    return a, sfa, b, sfb


def random_fp8_bytes(shape, seed: int = 0, allow_nan: bool = False) -> np.ndarray:
    """Uniform random e4m3fn bit patterns (exercises subnormals, -0 and, optionally, NaN)."""
    rng = np.random.default_rng(seed)
    q = rng.integers(0, 256, size=shape, dtype=np.uint8)
    if not allow_nan:
        q = np.where((q & 0x7F) == 0x7F, q & 0x80, q).astype(np.uint8)
    return q


def verify_isclose(output: np.ndarray, golden: np.ndarray, rtol: float, atol: float = 1e-9, error_tol: float = 1e-4):
    """Reference verifier restated (scripts/verify.py:14-35).  Returns (ok, ratio)."""
    o = _c(output, np.float32).reshape(-1); g = _c(golden, np.float32).reshape(-1)
    if o.size != g.size:
        return False, 1.0
    ratio = ctypes.c_double(0.0)
    lib().dga_oracle_verify_isclose(_p(o, c_f32p), _p(g, c_f32p), o.size, rtol, atol, ctypes.byref(ratio))
    return ratio.value <= error_tol, ratio.value


def ref_config(*args) -> list:
    """Run oracle/_ref/ref_config (the reference's own get_best_config.hpp compiled where it lies)."""
    exe = _HERE / "_ref" / "ref_config"
    if not exe.exists():
        raise FileNotFoundError(str(exe))
    out = subprocess.check_output([str(exe)] + [str(a) for a in args], text=True)
    return [int(x) for x in out.split()]


# --------------------------------------------------------------------------- tolerance

def abs_term_sum(a, sfa, b, sfb) -> np.ndarray:
    """S[m,n] = sum_kb |sfa*sfb| * sum_k |a||b|  (fp64): the magnitude the fp8 MFMA's internal
    alignment error scales with (see DESIGN.md 'Numerics of the fp8 MFMA datapath')."""
    tab = np.abs(np.nan_to_num(e4m3fn_table().astype(np.float64), nan=0.0))
    a = np.asarray(a, np.uint8); b = np.asarray(b, np.uint8)
    m, k = a.shape; n, _ = b.shape
    out = np.zeros((m, n))
    col_blk = np.arange(n) // 128
    for kb in range((k + 127) // 128):
        sl = slice(kb * 128, min(k, kb * 128 + 128))
        s = np.abs(np.asarray(sfa, np.float64)[:, kb][:, None] * np.asarray(sfb, np.float64)[col_blk, kb][None, :])
        out += s * (tab[a[:, sl]] @ tab[b[:, sl]].T)
    return out


def bf16_ulp_of(x: np.ndarray) -> np.ndarray:
    """Spacing of bf16 at |x| (8 significant bits; subnormal spacing 2^-133)."""
    ax = np.abs(np.asarray(x, np.float64))
    e = np.floor(np.log2(np.maximum(ax, 2.0 ** -126)))
    return 2.0 ** (e - 7)


# The parity bar (BASELINE.json north_star: "within 2 ULP bf16").  The fp8 MFMA does not accumulate
# its 128 products like an fp32 FMA chain: each octet of products is aligned to the octet's largest
# exponent and bits more than ~13 below it are dropped (measured: scripts/probe_mfma_numerics.py,
# gpurun_out of round 1; same behaviour DeepSeek-V3 reports for Hopper fp8 tensor cores).  For outputs
# that are not cancellation-dominated this is far below one bf16 ULP; for outputs near zero it is an
# absolute error proportional to S = abs_term_sum.  Hence:
#     |got - want| <= MAX_ULP * ulp_bf16(want) + MFMA_ALIGN_EPS * S
# MFMA_ALIGN_EPS = 2^-15 holds for amax-quantised data once a row has a full 128-wide block (worst measured 2^-17 at
# K >= 128): S then sums over enough products that one octet's dropped bits are small against it.  With K < 128 (a
# couple of octets) S is barely larger than the largest product and the ratio approaches the hardware's own envelope,
# 2^-12 (worst measured on K = 16: 2^-14.8) -- eps_for_k() returns that envelope there.
MAX_ULP = 2
MFMA_ALIGN_EPS = 2.0 ** -15
MFMA_ALIGN_EPS_HW = 2.0 ** -12


def eps_for_k(k: int) -> float:
    return MFMA_ALIGN_EPS if k >= 128 else MFMA_ALIGN_EPS_HW


def parity_excess(got_bits, want_bits, a, sfa, b, sfb, max_ulp: int = MAX_ULP, eps: float = MFMA_ALIGN_EPS):
    """Returns (ok, worst) where worst = max over elements of
    (|got-want| - max_ulp*ulp(want)) / S  (<= eps passes).  NaN positions must coincide."""
    g = bf16_bits_to_f32(got_bits).astype(np.float64)
    w = bf16_bits_to_f32(want_bits).astype(np.float64)
    gn, wn = np.isnan(g), np.isnan(w)
    if not np.array_equal(gn, wn):
        return False, float("inf")
    fin = ~wn
    if not fin.any():
        return True, 0.0
    s = abs_term_sum(a, sfa, b, sfb)
    diff = np.abs(np.where(fin, g - w, 0.0))
    allow = max_ulp * bf16_ulp_of(w)
    exc = np.where(fin, np.maximum(diff - allow, 0.0), 0.0)
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.where(s > 0, exc / s, np.where(exc > 0, np.inf, 0.0))
    worst = float(r.max(initial=0.0))
    return worst <= eps, worst


def parity_report(got_bits, want_bits, a, sfa, b, sfb) -> dict:
    d = bf16_ulp_diff(got_bits, want_bits)
    ok, worst = parity_excess(got_bits, want_bits, a, sfa, b, sfb, eps=float("inf"))
    n = max(1, d.size)
    return {"max_ulp": int(d.max(initial=0)), "frac_gt_max_ulp": float((d > MAX_ULP).sum()) / n,
            "worst_excess_over_S": worst, "nan_positions_equal": ok}


def assert_parity(got_bits, want_bits, a, sfa, b, sfb, eps: float = None, frac: float = 2e-3):
    """The parity bar used by every GPU test:
      (1) NaN positions identical;
      (2) every element: |got-want| <= 2 ulp_bf16(want) + eps * S      (eps = 2^-15 for amax-quantised
          data, 2^-12 = the hardware's worst-case envelope for arbitrary bit patterns);
      (3) at most `frac` of the elements (or 8 of them, on small samples) need the eps term at all
          (cancellation-dominated outputs);
          the reference's own verifier tolerates a 1e-4 mismatch fraction (scripts/verify.py:10-35)."""
    if eps is None:
        eps = eps_for_k(np.asarray(a).shape[-1])
    rep = parity_report(got_bits, want_bits, a, sfa, b, sfb)
    assert rep["nan_positions_equal"], "NaN positions differ"
    assert rep["worst_excess_over_S"] <= eps, f"excess error {rep['worst_excess_over_S']:.3e} * S > eps {eps:.3e}: {rep}"
    # (3) is a statement about a population: on a small sample allow a handful of elements whatever the fraction
    size = int(np.asarray(got_bits).size)
    assert rep["frac_gt_max_ulp"] * size <= max(frac * size, 8), \
        f"{rep['frac_gt_max_ulp']:.2e} of {size} elements beyond {MAX_ULP} ulp: {rep}"
    return rep
