"""Host-side mirror of the reference's operator API, over the C ABI (no torch types cross it).

Reference surface being mirrored (paths relative to /root/reference):
  * ``deep_gemm_ascend.run_mmad_rtc(x, y, z)`` / ``run_mmad_bench(x, y, z, params)``
    (deep_gemm_ascend/framework/deep_gemm_ascend/__init__.py:1-4,
    framework/csrc/python_api.cpp:18-35): void return, output written in place into the
    caller-allocated ``z``, current device stream.
  * the aclnn operator ``CatlassDynamicMatmul(self, mat2) -> out`` hooks
    (aclnn_catlass_dynamic_matmul/op_host/catlass_dynamic_matmul.cpp:16-80): infer_shape,
    infer_dtype, tiling.
  * new for the fp8 hot path (names from upstream DeepGEMM):
    ``gemm_fp8_fp8_bf16_nt((a, sfa), (b, sfb), out)`` and
    ``m_grouped_gemm_fp8_fp8_bf16_nt_masked((a, sfa), (b, sfb), out, masked_m, expected_m)``.

PyTorch is used only for device memory and the current stream.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import DGAError, Platform, Problem, Tiling

_FP8 = getattr(torch, "float8_e4m3fn", None)


# Host time per call matters for the short shapes of the reference's sweep list (a 64 x 32768 x 512 GEMM is 6 us of device
# time): the helpers below keep a call at a handful of Python-level operations -- no tensor views, no stream / device objects,
# no message formatting unless a check fails (scripts/host_overhead.py: 12.5 -> 5 us per call).
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream_of(index: int) -> int:
    """The current HIP stream of device `index` as the raw handle the C ABI takes."""
    if _raw_stream is not None:
        return _raw_stream(index)
    return torch.cuda.current_stream(index).cuda_stream


def _stream_ptr(t: torch.Tensor) -> int:
    if t.is_cuda:
        return _stream_of(t.device.index)
    return 0


def _fail(msg: str):
    # DGA_HOST_ASSERT analogue (csrc/utils/exception.hpp:26-33)
    raise DGAError(-2, "host assert", msg)


def _require(cond: bool, msg: str):
    if not cond:
        _fail(msg)


def _fp8_bytes(t: torch.Tensor) -> torch.Tensor:
    """Checks that `t` holds e4m3fn bytes; the tensor itself is returned (its data pointer is what the C ABI takes)."""
    if t.dtype != _FP8 and t.dtype != torch.uint8:
        _fail(f"fp8 operand must be float8_e4m3fn or uint8 bytes, got {t.dtype}")
    return t


_WORKSPACES = {}   # (kind, device index, stream handle) -> uint8 tensor
_RETIRED = []      # outgrown buffers: a HIP graph captured earlier may still hold their address, so they stay alive


def _scratch(kind: str, device, need: int, stream: Optional[int] = None) -> Tuple[Optional[int], int]:
    """Device scratch, one grow-only buffer per (kind, device, stream): two GEMMs issued on different streams never
    share split-K slabs or padded operand copies, and the capture stream of a HIP graph has a buffer of its own.  A
    buffer that has to grow is replaced, not freed (a captured graph may replay into the old one); growth is geometric
    so that few are ever retired.  The callee never allocates (the op runtime hands the workspace in)."""
    if need == 0:
        return None, 0
    dev = device if isinstance(device, torch.device) else torch.device(device)
    index = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (kind, index, _stream_of(index) if stream is None else stream)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < need:
        if buf is not None:
            _RETIRED.append(buf)
            need = max(need, buf.numel() * 3 // 2)
        buf = torch.empty((need,), dtype=torch.uint8, device=dev)
        _WORKSPACES[key] = buf
    return buf.data_ptr(), buf.numel()


def release_scratch(retired_only: bool = True) -> int:
    """Frees the outgrown scratch buffers (`retired_only`) or every scratch buffer of the process; returns the bytes released.
    Worst case held per (device, stream): the selector bounds split-K slabs to 256 MiB (dga_tiling.cpp kMaxSlabBytes) and the
    odd-K padding copies to the operands' size, and growth is geometric, so at most ~1.5x the largest workspace ever asked for
    on that stream -- plus the retired ones until this is called.  Only call it when no captured HIP graph that ran a GEMM
    through this module will be replayed again (a graph holds the address of the buffer it was captured with)."""
    n = sum(b.numel() for b in _RETIRED)
    _RETIRED.clear()
    if not retired_only:
        n += sum(b.numel() for b in _WORKSPACES.values())
        _WORKSPACES.clear()
    return int(n)


_WS_BYTES = {}     # bytes of a Tiling -> dga_workspace_bytes of it (pure function of the struct)


def _workspace(t: Tiling, device, stream: Optional[int] = None) -> Tuple[Optional[int], int]:
    """Split-K slabs / odd-K padding of the fp8 kernels, sized by dga_workspace_bytes."""
    key = bytes(t)
    need = _WS_BYTES.get(key)
    if need is None:
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
        need = _WS_BYTES[key] = workspace_bytes(t)
    return _scratch("fp8", device, need, stream)


def _mmad_workspace(batch, m, n, k, x) -> Tuple[Optional[int], int]:
    return _scratch("mmad", x.device, int(_lib.lib().dga_mmad_workspace_bytes(batch, m, n, k, x.data_ptr())))


POLICY_PLAIN, POLICY_PINGPONG, POLICY_CONTINUOUS, POLICY_STRICT, POLICY_LOADER_WAVES, POLICY_PERSISTENT = 0, 1, 2, 3, 4, 5
POLICY_CONTINUOUS_PERSISTENT = 6
POLICY_BF16_EXACT = 7

# The three arithmetic policies of the fp8 GEMMs (README.md "Numerics"; include/dga_hip.h, dispatchPolicyTag):
#   "fast"        the fp8 matrix instruction (whatever schedule the tiling names) -- the throughput form
#   "bf16_exact"  e4m3 -> bf16 in registers (exact), bf16 matrix instruction: exact products, fp32-class block sums
#   "strict"      fp32-input matrix instruction in the oracle's own order: bit-identical to the reference CPU path
#   "auto"        "bf16_exact" where it is nearly free -- the decode rows the one-launch workgroup split-K takes (the weights are
#                 streamed, the exact arithmetic rides along: +2..+15 %) -- and "fast" everywhere else; dense calls without an
#                 explicit tiling only.
# A call that names neither a policy nor a tiling runs $DGA_DEFAULT_POLICY, default "bf16_exact": the fastest policy whose outputs
# stay inside the operator's contract (within 2 bf16 ULP of the fp32-accumulate CPU path; <= 1e-5 of the outputs of a 4096^3
# problem differ by more, each a sum that cancels to the level of the fp32 reference's own rounding).  "fast" is an opt-in: the
# fp8 matrix instruction drops product bits ~13 below each octet's largest (6.7e-4 of the same outputs beyond 2 ULP).  An explicit
# tiling_ keeps the arithmetic its dispatchPolicyTag names.
#   "fast_ue8m0"  "fast" for scale tensors whose values are exact powers of two (UE8M0 scales: per_token_cast_to_fp8(...,
#                 use_ue8m0=True), upstream DeepGEMM's convention for hardware-scaled MFMAs): the scales ride in the matrix
#                 instruction's E8M0 operands and the MFMA accumulates in place -- no promotion on the vector pipe.  Same outputs
#                 as "fast" up to fp32 rounding order; a scale that is not a power of two is read as its exponent alone.
#   "bf16_exact_ue8m0"  "bf16_exact" for power-of-two scales: the scales are folded into the exact e4m3 -> bf16 conversions and the
#                 bf16 matrix instruction accumulates in place -- the in-contract arithmetic without its promotion.
ARITHMETIC_POLICIES = {"fast": None, "bf16_exact": POLICY_BF16_EXACT, "strict": POLICY_STRICT, "auto": None, "fast_ue8m0": None,
                       "bf16_exact_ue8m0": POLICY_BF16_EXACT | 16}
POLICY_UE8M0_SCALES = 16      # DGA_POLICY_UE8M0_SCALES: a flag beside the fast-path schedules


def default_policy() -> str:
    """$DGA_DEFAULT_POLICY as the C library parsed and validated it (dga_default_policy: once per process, the same answer for every
    front end); a value that names no policy raises instead of silently changing the arithmetic."""
    buf = ctypes.create_string_buffer(32)
    rc = _lib.lib().dga_default_policy(buf, 32)
    if rc:
        _fail(f"$DGA_DEFAULT_POLICY={os.environ.get('DGA_DEFAULT_POLICY')!r} names no arithmetic policy (one of {sorted(ARITHMETIC_POLICIES)})")
    return buf.value.decode()


class _DefaultPolicy:
    """Lazy `_DEFAULT_POLICY` (the library need not be loaded at import time): str(...) / == compare against the parsed name."""
    def __str__(self):
        return default_policy()

    def __eq__(self, other):
        return default_policy() == other

    def __hash__(self):
        return hash(default_policy())


_DEFAULT_POLICY = _DefaultPolicy()


def _with_policy(t: Tiling, strict: bool, policy: Optional[str] = None) -> Tiling:
    """A copy of the tiling with the arithmetic policy's dispatchPolicyTag (strict=True is policy="strict")."""
    _require(policy is None or policy in ARITHMETIC_POLICIES, f"policy must be one of {sorted(ARITHMETIC_POLICIES)}")
    _require(not (strict and policy not in (None, "strict")), "strict=True contradicts policy=%r" % (policy,))
    tag = POLICY_STRICT if strict else ARITHMETIC_POLICIES.get(policy)
    if policy == "fast_ue8m0" and not strict:
        _require((t.dispatchPolicyTag & 15) not in (POLICY_STRICT, POLICY_BF16_EXACT), "policy='fast_ue8m0' needs a fast-path tiling")
        tag = (t.dispatchPolicyTag & 15) | POLICY_UE8M0_SCALES
    if tag is None or t.dispatchPolicyTag == tag:
        return t
    c = Tiling()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(t), ctypes.sizeof(Tiling))
    c.dispatchPolicyTag = tag
    return c


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _device_guard(*ts: torch.Tensor):
    """All tensors on one HIP device; that device current for the call (nothing to do when it already is)."""
    dev = ts[0].device
    for t in ts:
        if t.device != dev:
            _fail("all tensors must live on one device")
    if dev.type != "cuda":
        _fail("tensors must be on a HIP device (there is no CPU path)")
    return _NO_GUARD if torch.cuda.current_device() == dev.index else torch.cuda.device(dev)


_PLANS = {}        # what a GEMM call without an explicit tiling resolves to: key -> Tiling (with the policy's tag)


def _plans_clear():
    _PLANS.clear()


def _planned(index: int, m: int, n: int, k: int, groups: int, expected_m: int, contiguous: bool, strict: bool,
             policy: Optional[str]) -> Tiling:
    """tiling(...) + the arithmetic policy's tag, remembered per problem: the C side's (m,n,k) cache answers the same question, but
    through two ctypes calls and a struct copy per GEMM.  Dropped whenever the cache file, the cache or the predictor changes."""
    if policy is None and not strict:
        policy = default_policy()
    key = (index, m, n, k, groups, expected_m, contiguous, strict, policy)
    t = _PLANS.get(key)
    if t is None:
        if len(_PLANS) > 4096:
            _PLANS.clear()
        _require(policy is None or policy in ARITHMETIC_POLICIES, f"policy (or $DGA_DEFAULT_POLICY) must be one of {sorted(ARITHMETIC_POLICIES)}")
        if policy == "auto":   # the exact arithmetic where the decode kernel carries it, the fast path elsewhere
            _require(not strict, "strict=True contradicts policy='auto'")
            tb = tiling(m, n, k, policy="bf16_exact") if groups == 1 and not contiguous else None
            if tb is not None and tb.kernelSerial == 6:   # DGA_KERNEL_SPLITK_WORKGROUP
                t = tb
            else:
                t = tiling(m, n, k, groups=groups, expected_m=expected_m, contiguous=contiguous)
            _PLANS[key] = t
            return t
        t = tiling(m, n, k, groups=groups, expected_m=expected_m, contiguous=contiguous,
                   policy="bf16_exact" if policy in ("bf16_exact", "bf16_exact_ue8m0") else None)
        t = _PLANS[key] = _with_policy(t, strict, policy)
    return t


# ----------------------------------------------------------------------------- operator hooks

def infer_shape(self_shape: Sequence[int], mat2_shape: Sequence[int]) -> Tuple[int, int]:
    """InferShape hook (catlass_dynamic_matmul.cpp:16-35)."""
    a = (ctypes.c_int64 * len(self_shape))(*self_shape)
    b = (ctypes.c_int64 * len(mat2_shape))(*mat2_shape)
    out = (ctypes.c_int64 * 2)()
    _lib.check(_lib.lib().dga_infer_shape(a, len(self_shape), b, len(mat2_shape), out), "infer_shape")
    return int(out[0]), int(out[1])


def infer_dtype(self_dtype: int, mat2_dtype: int) -> int:
    """InferDataType hook (catlass_dynamic_matmul.cpp:37-46)."""
    out = ctypes.c_int(0)
    _lib.check(_lib.lib().dga_infer_dtype(self_dtype, mat2_dtype, ctypes.byref(out)), "infer_dtype")
    return out.value


def _problem(m, n, k, groups=1, expected_m=0, dtype=_lib.DT_FP8_E4M3FN, contiguous=False) -> Problem:
    return Problem(m, n, k, groups, expected_m, _lib.LAYOUT_ROW_MAJOR, _lib.LAYOUT_COLUMN_MAJOR,
                   _lib.LAYOUT_ROW_MAJOR, dtype, _lib.PROBLEM_CONTIGUOUS_M if contiguous else 0)


def tiling(m: int, n: int, k: int, groups: int = 1, expected_m: int = 0, contiguous: bool = False,
           policy: Optional[str] = None) -> Tiling:
    """TilingFunc hook with the (m,n,k) cache (catlass_dynamic_matmul_tiling.cpp:77-122).
    contiguous=True: the contiguous-grouped layout (m = total rows, groups = number of B matrices).
    policy="bf16_exact": that policy's own tile / split-K pick (dga_tiling_bf16_exact); "strict": the tag alone."""
    t = Tiling()
    p = _problem(m, n, k, groups, expected_m, contiguous=contiguous)
    if policy == "bf16_exact":
        _lib.check(_lib.lib().dga_tiling_bf16_exact(ctypes.byref(p), ctypes.byref(t)), "tiling_bf16_exact")
        return t
    _lib.check(_lib.lib().dga_tiling(ctypes.byref(p), ctypes.byref(t)), "tiling")
    return _with_policy(t, False, policy) if policy else t


def tiling_check(t: Tiling) -> int:
    """dga_tiling_check: 0 if the compiled menu holds this tiling, else the (negative) status every fp8 GEMM entry returns for
    it before any launch (the TilingFunc's GRAPH_FAILED, catlass_dynamic_matmul_tiling.cpp:86-100)."""
    return int(_lib.lib().dga_tiling_check(ctypes.byref(t)))


def select_kernel(m: int, n: int, k: int, platform: Optional[Platform] = None, groups: int = 1,
                  expected_m: int = 0) -> Tiling:
    """SelectKernel without the cache (select_kernel.cpp:333-369); platform None = MI355X."""
    t = Tiling()
    p = _problem(m, n, k, groups, expected_m)
    pp = ctypes.byref(platform) if platform is not None else None
    _lib.check(_lib.lib().dga_select_kernel(ctypes.byref(p), pp, ctypes.byref(t)), "select_kernel")
    return t


def predictor_load(path: Optional[str] = None) -> None:
    """Load a predictor weights file (None = tuned/predictor_mi355x.txt next to the library)."""
    _plans_clear()
    _lib.check(_lib.lib().dga_predictor_load(path.encode() if path else None), "predictor_load")


def predictor_unload() -> None:
    _plans_clear()
    _lib.lib().dga_predictor_unload()


def predictor_loaded() -> bool:
    return bool(_lib.lib().dga_predictor_loaded())


def predict_time_us(m: int, n: int, k: int, t: Tiling) -> float:
    """The model's time for running (m,n,k) with tiling t (TilingPredictor.predict_batch for one row)."""
    us = ctypes.c_float(0)
    p = _problem(m, n, k)
    _lib.check(_lib.lib().dga_predict_time_us(ctypes.byref(p), ctypes.byref(t), ctypes.byref(us)), "predict_time_us")
    return us.value


SELECTION_METHODS = {"greedy": 0, "topk_median": 1, "topk_dbscan": 2}   # get_best_config.py:431-525


def select_kernel_with_predictor(m: int, n: int, k: int, method: str = "greedy", topk: int = 10):
    """SelectKernelWithPredictor: (tiling, predicted_us, native_us); falls back to the heuristic tiling when the model
    is absent, the candidate list is short or the promised gain is below the threshold (get_best_config.py:587-621).  `method` /
    `topk`: the reference predictor's selection strategy (select_tiling_strategy, get_best_config.py:431-525)."""
    _require(method in SELECTION_METHODS, f"method: one of {sorted(SELECTION_METHODS)}")
    t = Tiling()
    p = _problem(m, n, k)
    a, b = ctypes.c_float(0), ctypes.c_float(0)
    _lib.check(_lib.lib().dga_select_kernel_with_predictor_ex(ctypes.byref(p), ctypes.byref(t), ctypes.byref(a), ctypes.byref(b),
                                                              SELECTION_METHODS[method], int(topk)), "select_kernel_with_predictor")
    return t, a.value, b.value


def select_tiling_strategy(preds, tiles, method: str = "greedy", topk: int = 10, dbscan_eps: float = 0.8, dbscan_min_samples: int = 2,
                           random_state: int = 0):
    """select_tiling_strategy (get_best_config.py:431-525) on predicted times `preds` [count] and tile triples `tiles` [count][3]
    (mTile, nTile, kTile): returns (picked index, members of the winning cluster -- topk_dbscan only, else []).  The reference draws a
    random member of that cluster; this library takes its fastest (random_state = 0) or member random_state % size."""
    import numpy as np
    _require(method in SELECTION_METHODS, f"method: one of {sorted(SELECTION_METHODS)}")
    pr = np.ascontiguousarray(preds, dtype=np.float32)
    tl = np.ascontiguousarray(tiles, dtype=np.int32).reshape(-1, 3) if tiles is not None else np.zeros((pr.size, 3), np.int32)
    _require(tl.shape[0] == pr.size, "one tile triple per prediction")
    picked, cnt = ctypes.c_int(-1), ctypes.c_int(0)
    members = (ctypes.c_int * max(1, pr.size))()
    _lib.check(_lib.lib().dga_select_tiling_strategy(pr.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                                     tl.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), int(pr.size),
                                                     SELECTION_METHODS[method], int(topk), float(dbscan_eps), int(dbscan_min_samples),
                                                     int(random_state), ctypes.byref(picked), members, ctypes.byref(cnt)),
               "select_tiling_strategy")
    return picked.value, [members[i] for i in range(cnt.value)]


def platform_mi355x() -> Platform:
    p = Platform()
    _lib.lib().dga_platform_mi355x(ctypes.byref(p))
    return p


def platform_ascend910b(core_num: int = 24) -> Platform:
    p = Platform()
    _lib.lib().dga_platform_ascend910b(ctypes.byref(p), core_num)
    return p


def workspace_bytes(t: Tiling) -> int:
    return int(_lib.lib().dga_workspace_bytes(ctypes.byref(t)))


def tiling_cache_open(path: Optional[str]):
    _plans_clear()
    _lib.check(_lib.lib().dga_tiling_cache_open(path.encode() if path else None), "tiling_cache_open")


def tiling_cache_clear():
    _plans_clear()
    _lib.lib().dga_tiling_cache_clear()


def tiling_cache_size() -> int:
    return int(_lib.lib().dga_tiling_cache_size())


# ----------------------------------------------------------------------------- 28-int Config

def get_best_config(batch: int, m: int, n: int, k: int) -> list:
    out = (ctypes.c_uint32 * 28)()
    _lib.check(_lib.lib().dga_get_best_config(batch, m, n, k, out), "get_best_config")
    return list(out)


def get_bench_config(m, n, k, m_sections, n_sections, m_sec_o_blocks, n_sec_o_blocks, k_o_iter_blocks,
                     db_o_blocks) -> list:
    out = (ctypes.c_uint32 * 28)()
    _lib.check(_lib.lib().dga_get_bench_config(m, n, k, m_sections, n_sections, m_sec_o_blocks, n_sec_o_blocks,
                                               k_o_iter_blocks, db_o_blocks, out), "get_bench_config")
    return list(out)


CONFIG_FIELDS = ("k_iters batch m n k m_sections n_sections m_blocks n_blocks k_blocks m_sc_blocks n_sc_blocks "
                 "m_sec_o_blocks n_sec_o_blocks k_o_iter_blocks db_o_blocks m_o_fix n_o_fix k_o_fix db_o_num "
                 "m_parts n_parts r_m_parts r_n_parts r_m_blocks r_n_blocks r_k_blocks r_db_num").split()


def bbit_params(m, n, k, m_sections, n_sections, m_sec_o_blocks, n_sec_o_blocks, k_o_iter_blocks, db_o_blocks):
    out = (ctypes.c_uint32 * 28)()
    _lib.check(_lib.lib().dga_bbit_params(m, n, k, m_sections, n_sections, m_sec_o_blocks, n_sec_o_blocks,
                                          k_o_iter_blocks, db_o_blocks, out), "bbit_params")
    return list(out)


def bench_params_fill(m: int, n: int, k: int, params6: Sequence[int]) -> list:
    arr = (ctypes.c_int32 * 28)(*list(params6)[:6], *([0] * 22))
    _lib.check(_lib.lib().dga_bench_params_fill(m, n, k, arr), "bench_params_fill")
    return list(arr)


# ----------------------------------------------------------------------------- the hot path

def gemm_fp8_fp8_bf16_nt(lhs: Tuple[torch.Tensor, torch.Tensor], rhs: Tuple[torch.Tensor, torch.Tensor],
                         out: torch.Tensor, tiling_: Optional[Tiling] = None, sync: bool = False,
                         strict: bool = False, policy: Optional[str] = None,
                         zero_padded: Optional[Tuple[bool, bool]] = None) -> None:
    """out[M,N] (bf16, written in place) = (A[M,K] fp8, sfa[M,ceil(K/128)]) x (B[N,K] fp8, sfb[ceil(N/128),ceil(K/128)])^T.

    A and B may be row-strided views (unit inner stride; a row stride that is K or a multiple of 16 bytes): rows that start on
    16-byte boundaries are read where they lie.  For K % 16 != 0 that also needs zeros from byte K to the next 16-byte boundary
    of each row -- true of what per_token_cast_to_fp8 / per_block_cast_to_fp8(..., aligned_rows=True) return (they mark their
    results), or promised by the caller with zero_padded=(a_is, b_is); an operand without the promise is re-laid out by the
    padding pass, alone (dga_gemm_fp8_fp8_bf16_nt_strided).

    strict=True (= policy="strict") runs the exact-arithmetic kernel (dispatchPolicyTag 3): fp32 products and sums in the
    reference CPU path's own order, bit-identical to the oracle, at the fp32 matrix rate.  policy="bf16_exact"
    (dispatchPolicyTag 7) up-converts the bytes to bf16 in registers and sums on the bf16 matrix instruction: exact products,
    fp32-class sums, about half the fast path's rate.  The default fp8-MFMA path ("fast") differs from both on
    cancellation-dominated outputs (README.md, "Numerics").

    Asynchronous on the current stream (the reference syncs on every call, gemm.hpp:110;
    pass sync=True for that behaviour)."""
    a, sfa = lhs
    b, sfb = rhs
    _fp8_bytes(a); _fp8_bytes(b)
    if a.dim() != 2 or b.dim() != 2 or out.dim() != 2:
        _fail("rank must be 2")
    m, k = a.shape
    n, k2 = b.shape
    if k != k2:
        _fail("self dimk is not equal with mat2 dimk")
    if out.shape[0] != m or out.shape[1] != n:
        _fail(f"out must be [{m},{n}]")
    if out.dtype != torch.bfloat16:
        _fail("out must be bfloat16")
    kb, nb = (k + 127) // 128, (n + 127) // 128
    if sfa.dtype != torch.float32 or sfb.dtype != torch.float32:
        _fail("scales must be float32")
    if sfa.dim() != 2 or sfa.shape[0] != m or sfa.shape[1] != kb:
        _fail(f"sfa must be [{m},{kb}]")
    if sfb.dim() != 2 or sfb.shape[0] != nb or sfb.shape[1] != kb:
        _fail(f"sfb must be [{nb},{kb}]")
    if not (sfa.is_contiguous() and sfb.is_contiguous() and out.is_contiguous()):
        _fail("scales and out must be contiguous")
    if (k > 1 and (a.stride(1) != 1 or b.stride(1) != 1)) or (m > 1 and a.stride(0) < k) or (n > 1 and b.stride(0) < k):
        _fail("operands must be row-major with unit inner stride (row-strided views are accepted)")
    lda = a.stride(0) if m > 1 else k
    ldb = b.stride(0) if n > 1 else k
    strided = lda != k or ldb != k
    if zero_padded is None:
        zero_padded = (bool(getattr(a, "_dga_zero_padded", False)), bool(getattr(b, "_dga_zero_padded", False)))
    with _device_guard(a, b, sfa, sfb, out):
        index = out.device.index
        if tiling_ is None:
            tiling_ = _planned(index, m, n, k, 1, 0, False, strict, policy)
        else:
            tiling_ = _with_policy(tiling_, strict, policy)
        stream = _stream_of(index)
        ws_ptr, ws_bytes = _workspace(tiling_, out.device, stream)
        if strided:
            flags = (_lib.ROWS_A_ZERO_PADDED if zero_padded[0] else 0) | (_lib.ROWS_B_ZERO_PADDED if zero_padded[1] else 0)
            rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt_strided(a.data_ptr(), lda, sfa.data_ptr(), b.data_ptr(), ldb, sfb.data_ptr(),
                                                             out.data_ptr(), m, n, k, flags, ctypes.byref(tiling_), ws_ptr,
                                                             ws_bytes, stream)
        else:
            rc = _lib.lib().dga_gemm_fp8_fp8_bf16_nt(a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(),
                                                     out.data_ptr(), m, n, k, ctypes.byref(tiling_), ws_ptr, ws_bytes, stream)
        if rc:
            _lib.check(rc, "gemm_fp8_fp8_bf16_nt")
        if sync:
            torch.cuda.current_stream(out.device).synchronize()


def gemm_fp8_loop_clock(lhs, rhs, out: torch.Tensor, tiling_: Optional[Tiling] = None, launches: int = 50):
    """(clock_mhz, loop_us): the shader clock the chip holds inside the main loop of the dense kernel `tiling_` selects,
    and that loop's duration, from the loop-clock build (dga_gemm_fp8_loop_clock; a diagnostic: it synchronises)."""
    a, sfa = lhs
    b, sfb = rhs
    _fp8_bytes(a); _fp8_bytes(b)
    m, k = a.shape
    n = b.shape[0]
    with _device_guard(a, b, sfa, sfb, out):
        if tiling_ is None:
            tiling_ = tiling(m, n, k)
        tiles = -(-m // max(1, tiling_.m1)) * -(-n // max(1, tiling_.n1))
        ws_ptr, ws_bytes = _scratch("clock", out.device, tiles * 128)
        mhz, us = ctypes.c_float(0), ctypes.c_float(0)
        rc = _lib.lib().dga_gemm_fp8_loop_clock(a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), out.data_ptr(),
                                                m, n, k, ctypes.byref(tiling_), ws_ptr, ws_bytes, int(launches),
                                                _stream_ptr(out), ctypes.byref(mhz), ctypes.byref(us))
        _lib.check(rc, "gemm_fp8_loop_clock")
    return mhz.value, us.value


def mfma_ceiling(policy: str = "fast", launches: int = 300, device=None) -> float:
    """TFLOP/s the matrix pipe of this device sustains on the policy's inner step with the operands already in registers
    (dga_mfma_ceiling: matrix instruction + fp32 promotion [+ in-register conversions], two waves per SIMD on every CU,
    random e4m3 bytes, the last of `launches` back-to-back launches).  A diagnostic: it synchronises."""
    _require(policy in ("fast", "bf16_exact"), "mfma_ceiling: policy must be 'fast' or 'bf16_exact'")
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    with torch.cuda.device(dev):
        cus = torch.cuda.get_device_properties(dev).multi_processor_count
        ws_ptr, ws_bytes = _scratch("ceiling", dev, 16384 + 2048 * cus)
        tf = ctypes.c_float(0)
        rc = _lib.lib().dga_mfma_ceiling(0 if policy == "fast" else 1, int(launches), ws_ptr, ws_bytes,
                                         torch.cuda.current_stream(dev).cuda_stream, ctypes.byref(tf))
        _lib.check(rc, "mfma_ceiling")
    return tf.value


def m_grouped_gemm_fp8_fp8_bf16_nt_masked(lhs, rhs, out: torch.Tensor, masked_m: torch.Tensor, expected_m: int,
                                          tiling_: Optional[Tiling] = None, sync: bool = False,
                                          strict: bool = False, policy: Optional[str] = None) -> None:
    """Grouped masked-M GEMM: a [G,Mmax,K], sfa [G,Mmax,KB], b [G,N,K], sfb [G,NB,KB], out [G,Mmax,N] bf16;
    only rows < masked_m[g] of out[g] are written (upstream DeepGEMM's convention)."""
    a, sfa = lhs
    b, sfb = rhs
    _fp8_bytes(a); _fp8_bytes(b)
    if a.dim() != 3 or b.dim() != 3 or out.dim() != 3:
        _fail("rank must be 3")
    g, mmax, k = a.shape
    g2, n, k2 = b.shape
    if g != g2 or k != k2:
        _fail("group / k mismatch")
    if tuple(out.shape) != (g, mmax, n) or out.dtype != torch.bfloat16:
        _fail("out must be [G,Mmax,N] bfloat16")
    kb, nb = (k + 127) // 128, (n + 127) // 128
    if tuple(sfa.shape) != (g, mmax, kb) or sfa.dtype != torch.float32:
        _fail(f"sfa must be [{g},{mmax},{kb}] f32")
    if tuple(sfb.shape) != (g, nb, kb) or sfb.dtype != torch.float32:
        _fail(f"sfb must be [{g},{nb},{kb}] f32")
    if masked_m.dtype != torch.int32 or masked_m.dim() != 1 or masked_m.shape[0] != g:
        _fail("masked_m must be int32 [G]")
    if not (a.is_contiguous() and b.is_contiguous() and sfa.is_contiguous() and sfb.is_contiguous() and out.is_contiguous() and
            masked_m.is_contiguous()):
        _fail("operands must be contiguous")
    expected_m = int(expected_m)
    with _device_guard(a, b, sfa, sfb, out, masked_m):
        index = out.device.index
        if tiling_ is None:
            tiling_ = _planned(index, mmax, n, k, g, expected_m, False, strict, policy)
        else:
            tiling_ = _with_policy(tiling_, strict, policy)
        stream = _stream_of(index)
        ws_ptr, ws_bytes = _workspace(tiling_, out.device, stream)
        rc = _lib.lib().dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked(
            a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), out.data_ptr(), masked_m.data_ptr(),
            g, mmax, n, k, expected_m, ctypes.byref(tiling_), ws_ptr, ws_bytes, stream)
        if rc:
            _lib.check(rc, "m_grouped_gemm_fp8_fp8_bf16_nt_masked")
        if sync:
            torch.cuda.current_stream(out.device).synchronize()


def m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(a_rows: torch.Tensor, sfa_src: torch.Tensor, sfa_byte_offset: int,
                                                  sfa_ld: int, rhs, out_rows: torch.Tensor, row_index: torch.Tensor,
                                                  masked_m: torch.Tensor, m_max: int, expected_m: int = 0,
                                                  tiling_: Optional[Tiling] = None, sync: bool = False,
                                                  strict: bool = False, policy: Optional[str] = None) -> None:
    """Masked grouped GEMM on rows that stay where they are (dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed):
    row r of group g is row row_index[g * m_max + r] of the flat byte rows `a_rows` [rows, lda] (first K bytes = fp8),
    its 1x128 scales start at byte sfa_byte_offset of row row_index[...] of `sfa_src` viewed with sfa_ld floats per row
    (the same payload rows, or a separate [rows, KB] float tensor with offset 0), and its result is written to that row
    of `out_rows` [rows, ldc] bf16.  b [G,N,K], sfb [G,NB,KB]; only r < masked_m[g] is read or written."""
    b, sfb = rhs
    _fp8_bytes(b); _fp8_bytes(a_rows)
    _require(a_rows.dim() == 2 and a_rows.stride(1) == 1 and out_rows.dim() == 2 and out_rows.stride(1) == 1, "2-D row tensors")
    g, n, k = b.shape
    rows, lda = a_rows.shape[0], a_rows.stride(0)
    kb, nb = (k + 127) // 128, (n + 127) // 128
    _require(a_rows.shape[1] >= k and out_rows.shape[1] >= n and out_rows.dtype == torch.bfloat16, "row widths")
    _require(out_rows.shape[0] >= rows, "out_rows must have a row for every source row")
    _require(tuple(sfb.shape) == (g, nb, kb) and sfb.dtype == torch.float32, f"sfb must be [{g},{nb},{kb}] f32")
    _require(row_index.dtype == torch.int64 and row_index.numel() >= g * m_max and row_index.is_contiguous(), "row_index int64[G*m_max]")
    _require(masked_m.dtype == torch.int32 and tuple(masked_m.shape) == (g,), "masked_m must be int32 [G]")
    _require(sfa_byte_offset % 4 == 0 and sfa_ld >= kb, "scale rows must be float-aligned")
    # the kernel reads row r's scales at sfa_src + offset + r * sfa_ld floats: the tensor has to be what that arithmetic assumes
    _require(sfa_src.dim() == 2 and sfa_src.stride(1) == 1 and sfa_src.dtype in (torch.float32, torch.uint8),
             "sfa_src must be a 2-D float32 (or uint8 payload) row tensor with unit inner stride")
    _require(sfa_src.stride(0) * sfa_src.element_size() == 4 * sfa_ld, "sfa_ld must be sfa_src's row stride in floats")
    _require(sfa_src.data_ptr() % 4 == 0 and sfa_src.shape[0] >= rows and
             sfa_byte_offset + 4 * kb <= sfa_src.shape[1] * sfa_src.element_size(),
             "sfa_src must hold ceil(K/128) floats at sfa_byte_offset of each of the source's rows")
    with _device_guard(a_rows, b, sfb, out_rows, row_index, masked_m, sfa_src):
        if tiling_ is None:
            tiling_ = _planned(out_rows.device.index, m_max, n, k, g, int(expected_m), False, strict, policy)
        else:
            tiling_ = _with_policy(tiling_, strict, policy)
        rc = _lib.lib().dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(
            a_rows.data_ptr(), lda, sfa_src.data_ptr() + sfa_byte_offset, sfa_ld, b.data_ptr(), sfb.data_ptr(),
            out_rows.data_ptr(), out_rows.stride(0), row_index.data_ptr(), rows, masked_m.data_ptr(), g, m_max, n, k,
            int(expected_m), ctypes.byref(tiling_), None, 0, _stream_ptr(out_rows))
        _lib.check(rc, "m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed")
        if sync:
            torch.cuda.current_stream(out_rows.device).synchronize()


def catlass_dynamic_matmul(self_: torch.Tensor, mat2: torch.Tensor, out: torch.Tensor, sync: bool = False) -> None:
    """The aclnn operator CatlassDynamicMatmul in its own dtypes: out[M,N] = self[M,K] @ mat2[K,N], all fp16 or all bf16.
    mat2 is the logical [K,N] matrix in column-major storage, i.e. the transposed view of a contiguous [N,K] tensor
    (`b.t()`), the operator's only layout (catlass_dynamic_matmul_tiling.cpp:83-84)."""
    _require(self_.dim() == 2 and mat2.dim() == 2 and out.dim() == 2, "rank must be 2")
    m, k = self_.shape
    k2, n = mat2.shape
    _require(k == k2, "self dimk is not equal with mat2 dimk")
    _require(self_.dtype in (torch.float16, torch.bfloat16) and mat2.dtype == self_.dtype and out.dtype == self_.dtype,
             "self, mat2 and out must share one 16-bit dtype")
    _require(tuple(out.shape) == (m, n) and out.is_contiguous() and self_.is_contiguous(), "self / out must be contiguous")
    _require(n == 0 or k == 0 or (mat2.stride(0) == 1 and mat2.stride(1) == k) or (k == 1 and mat2.stride(1) == 1) or
             (n == 1 and mat2.stride(0) == 1), "mat2 must be the transposed view of a contiguous [N,K] tensor (NT)")
    dt = _lib.DT_BF16 if self_.dtype == torch.bfloat16 else _lib.DT_FP16
    with _device_guard(self_, mat2, out):
        need = int(_lib.lib().dga_catlass_dynamic_matmul_workspace_bytes(m, n, k, self_.data_ptr(), mat2.data_ptr()))
        ws_ptr, ws_bytes = _scratch("op16", out.device, need)
        rc = _lib.lib().dga_catlass_dynamic_matmul(self_.data_ptr(), mat2.data_ptr(), out.data_ptr(), m, n, k, dt,
                                                   ws_ptr, ws_bytes, _stream_ptr(out))
        _lib.check(rc, "catlass_dynamic_matmul")
        if sync:
            torch.cuda.current_stream(out.device).synchronize()


def get_m_alignment_for_contiguous_layout() -> int:
    """Row alignment of the group segments in the contiguous-grouped layout (upstream DeepGEMM's name)."""
    return _lib.CONTIGUOUS_M_ALIGNMENT


def m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(lhs, rhs, out: torch.Tensor, m_indices: torch.Tensor,
                                              tiling_: Optional[Tiling] = None, sync: bool = False,
                                              strict: bool = False, policy: Optional[str] = None) -> None:
    """Contiguous-grouped GEMM (the prefill-side MoE layout): a [Msum,K], sfa [Msum,KB], b [G,N,K], sfb [G,NB,KB],
    out [Msum,N] bf16, m_indices int32 [Msum].  Row r is multiplied with b[m_indices[r]]; rows with a negative index
    are padding and stay untouched.  Group segments start at multiples of get_m_alignment_for_contiguous_layout()
    rows, padding rows follow a segment's valid rows."""
    a, sfa = lhs
    b, sfb = rhs
    _fp8_bytes(a); _fp8_bytes(b)
    _require(a.dim() == 2 and b.dim() == 3 and out.dim() == 2, "a/out rank 2, b rank 3")
    msum, k = a.shape
    g, n, k2 = b.shape
    _require(k == k2, "k mismatch")
    _require(tuple(out.shape) == (msum, n) and out.dtype == torch.bfloat16, "out must be [Msum,N] bfloat16")
    kb, nb = (k + 127) // 128, (n + 127) // 128
    _require(tuple(sfa.shape) == (msum, kb) and sfa.dtype == torch.float32, f"sfa must be [{msum},{kb}] f32")
    _require(tuple(sfb.shape) == (g, nb, kb) and sfb.dtype == torch.float32, f"sfb must be [{g},{nb},{kb}] f32")
    _require(m_indices.dtype == torch.int32 and tuple(m_indices.shape) == (msum,), "m_indices must be int32 [Msum]")
    for t in (a, b, sfa, sfb, out, m_indices):
        _require(t.is_contiguous(), "operands must be contiguous")
    with _device_guard(a, b, sfa, sfb, out, m_indices):
        if tiling_ is None:   # (the C side buckets Msum in its cache key and re-derives the workgroup count per call: not memoised here)
            if policy is None and not strict:
                policy = default_policy()      # no tiling, no policy: the operator's default arithmetic, as in the other entries
            _require(policy != "auto" or not strict, "strict=True contradicts policy='auto'")
            policy = "fast" if policy == "auto" else policy
            tiling_ = tiling(msum, n, k, groups=g, contiguous=True, policy="bf16_exact" if policy in ("bf16_exact", "bf16_exact_ue8m0") else None)
        tiling_ = _with_policy(tiling_, strict, policy)
        ws_ptr, ws_bytes = _workspace(tiling_, out.device)
        rc = _lib.lib().dga_m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(
            a.data_ptr(), sfa.data_ptr(), b.data_ptr(), sfb.data_ptr(), out.data_ptr(), m_indices.data_ptr(),
            msum, g, n, k, ctypes.byref(tiling_), ws_ptr, ws_bytes, _stream_ptr(out))
        _lib.check(rc, "m_grouped_gemm_fp8_fp8_bf16_nt_contiguous")
        if sync:
            torch.cuda.current_stream(out.device).synchronize()


_CAST_DT = {torch.float32: _lib.DT_FP32, torch.bfloat16: _lib.DT_BF16, torch.float16: _lib.DT_FP16}


def _cast(fn_name: str, x: torch.Tensor, block_rows: int, aligned_rows: bool = False, use_ue8m0: bool = False):
    _require(x.dim() == 2 and x.is_contiguous(), "x must be a contiguous [rows, k] tensor")
    _require(x.dtype in _CAST_DT, "x must be float32, bfloat16 or float16")
    rows, k = x.shape
    ldq = (k + 127) // 128 * 128 if aligned_rows else k   # whole 128-byte lines: a row's k blocks never straddle two
    q = torch.empty((rows, ldq), dtype=torch.uint8, device=x.device)
    sf = torch.empty(((rows + block_rows - 1) // block_rows, (k + 127) // 128), dtype=torch.float32, device=x.device)
    with _device_guard(x):
        if use_ue8m0:     # block scales rounded up to powers of two (dga_cast_to_fp8_*_ex, DGA_CAST_UE8M0)
            rc = getattr(_lib.lib(), fn_name + "_ex")(x.data_ptr(), _CAST_DT[x.dtype], rows, k, q.data_ptr(), ldq, sf.data_ptr(),
                                                      _lib.CAST_UE8M0, _stream_ptr(x))
        elif ldq != k:
            rc = getattr(_lib.lib(), fn_name + "_ld")(x.data_ptr(), _CAST_DT[x.dtype], rows, k, q.data_ptr(), ldq, sf.data_ptr(),
                                                      _stream_ptr(x))
        else:
            rc = getattr(_lib.lib(), fn_name)(x.data_ptr(), _CAST_DT[x.dtype], rows, k, q.data_ptr(), sf.data_ptr(),
                                              _stream_ptr(x))
        _lib.check(rc, fn_name)
    q = q.view(torch.float8_e4m3fn)
    if ldq != k:
        q = q[:, :k]
        q._dga_zero_padded = True   # (a Python attribute: views and copies made from it do not carry the promise)
    return q, sf


def per_token_cast_to_fp8(x: torch.Tensor, aligned_rows: bool = False, use_ue8m0: bool = False):
    """Activation quantiser: x [rows,k] -> (e4m3fn [rows,k], fp32 scales [rows, ceil(k/128)]), one scale per 1x128
    block: scale = amax/448, q = RNE-satfinite(x/scale) (the A-operand format of gemm_fp8_fp8_bf16_nt).
    aligned_rows=True: the result is a [rows, k] view of rows round_up(k, 128) bytes apart with zero tails, which
    gemm_fp8_fp8_bf16_nt reads in place whatever k is (no padding pass for k % 16 != 0; 1279 x 5003 x 7681: 80.6 -> 65.8 us.
    Rows only 16-byte aligned are read in place too but gain nothing: a 128-byte row piece that straddles two cache lines
    costs two requests on every re-read, profiles/r04_odd_k_rows.txt).
    use_ue8m0=True (upstream DeepGEMM's keyword): scale = 2^ceil(log2(amax / 448)), a power of two -- operands quantised this way
    on BOTH sides may be multiplied under policy="fast_ue8m0" (the scales ride in the matrix instruction's E8M0 operands)."""
    return _cast("dga_cast_to_fp8_1x128", x, 1, aligned_rows, use_ue8m0)


def per_block_cast_to_fp8(x: torch.Tensor, aligned_rows: bool = False, use_ue8m0: bool = False):
    """Weight quantiser: x [rows,k] -> (e4m3fn [rows,k], fp32 scales [ceil(rows/128), ceil(k/128)]), one scale per
    128x128 block (the B-operand format).  aligned_rows, use_ue8m0: as per_token_cast_to_fp8."""
    return _cast("dga_cast_to_fp8_128x128", x, 128, aligned_rows, use_ue8m0)


def route_tokens(expert_ids: torch.Tensor, groups: int):
    """(counts int64 [groups], pos int64 [T]): pos[t] = slot of token t in the expert-sorted order (dga_route_tokens)."""
    _require(expert_ids.dtype == torch.int64 and expert_ids.dim() == 1 and expert_ids.is_contiguous(), "expert_ids int64 [T]")
    counts = torch.empty((groups,), dtype=torch.int64, device=expert_ids.device)
    pos = torch.empty((expert_ids.numel(),), dtype=torch.int64, device=expert_ids.device)
    with _device_guard(expert_ids):
        rc = _lib.lib().dga_route_tokens(expert_ids.data_ptr(), expert_ids.numel(), groups, counts.data_ptr(), pos.data_ptr(),
                                         _stream_ptr(expert_ids))
        _lib.check(rc, "route_tokens")
    return counts, pos


def route_slots(keys: torch.Tensor, key_stride_bytes: int, rows: int, buckets: int, cap: int, counts: torch.Tensor,
                dest: torch.Tensor, overflow: torch.Tensor, key_div: int = 1, key_sub: int = 0, key_mul: int = 1,
                zero_counts: bool = True, tags: Optional[torch.Tensor] = None, tag_stride_bytes: int = 0,
                keys_byte_offset: int = 0, tags_byte_offset: int = 0, inverse: Optional[torch.Tensor] = None,
                inverse_base: int = 0) -> None:
    """Capacity-bounded slot assignment on the device (dga_route_slots): dest[r] = bucket(key_r) * cap + next free slot,
    -1 for unused rows and for rows of a full bucket (which also raises the sticky device flag `overflow`);
    inverse[dest[r]] = r + inverse_base (the slot -> row table of the indexed grouped GEMM)."""
    if inverse is not None:
        _require(inverse.dtype == torch.int64 and inverse.numel() >= buckets * cap and inverse.is_contiguous(),
                 "inverse int64[buckets * cap]")
    _require(counts.dtype == torch.int32 and counts.numel() >= buckets and counts.is_contiguous(), "counts int32[buckets]")
    _require(dest.dtype == torch.int64 and dest.numel() >= rows and dest.is_contiguous(), "dest int64[rows]")
    _require(overflow.dtype == torch.int32 and overflow.numel() >= 1, "overflow int32[1]")
    with _device_guard(keys, counts, dest, overflow):
        rc = _lib.lib().dga_route_slots(keys.data_ptr() + keys_byte_offset, key_stride_bytes, rows, key_div, key_sub, key_mul,
                                        buckets, cap, counts.data_ptr(), 1 if zero_counts else 0, dest.data_ptr(),
                                        (tags.data_ptr() + tags_byte_offset) if tags is not None else None,
                                        tag_stride_bytes, overflow.data_ptr(),
                                        inverse.data_ptr() if inverse is not None else None, inverse_base, _stream_ptr(dest))
        _lib.check(rc, "route_slots")


def copy_rows(dst: torch.Tensor, src: torch.Tensor, dst_index: Optional[torch.Tensor] = None,
              src_index: Optional[torch.Tensor] = None, rows: Optional[int] = None, row_bytes: Optional[int] = None,
              dst_byte_offset: int = 0, src_byte_offset: int = 0) -> None:
    """Indexed row copy between 2-D (row-strided) device tensors viewed as bytes (dga_copy_rows)."""
    _require(dst.dim() == 2 and src.dim() == 2 and dst.stride(1) == 1 and src.stride(1) == 1, "2-D row tensors")
    n = rows if rows is not None else (dst_index.numel() if dst_index is not None else
                                       src_index.numel() if src_index is not None else src.shape[0])
    rb = row_bytes if row_bytes is not None else min(dst.shape[1] * dst.element_size(), src.shape[1] * src.element_size())
    for ix in (dst_index, src_index):
        if ix is not None:
            _require(ix.dtype == torch.int64 and ix.is_contiguous() and ix.numel() >= n, "index must be int64[rows]")
    with _device_guard(dst, src):
        rc = _lib.lib().dga_copy_rows(dst.data_ptr() + dst_byte_offset, dst.stride(0) * dst.element_size(),
                                      dst_index.data_ptr() if dst_index is not None else None,
                                      src.data_ptr() + src_byte_offset, src.stride(0) * src.element_size(),
                                      src_index.data_ptr() if src_index is not None else None, rb, n, _stream_ptr(dst))
        _lib.check(rc, "copy_rows")


def copy_rows2(dst0, src0, bytes0, dst1, src1, bytes1, dst_index=None, src_index=None, rows=None,
               dst0_off=0, src0_off=0, dst1_off=0, src1_off=0) -> None:
    """Two indexed row copies with shared indices in one launch (dga_copy_rows2); tensors are 2-D row tensors viewed as
    bytes, *_off are byte offsets inside a row."""
    for t in (dst0, src0, dst1, src1):
        _require(t.dim() == 2 and t.stride(1) == 1, "2-D row tensors")
    n = rows if rows is not None else (dst_index.numel() if dst_index is not None else
                                       src_index.numel() if src_index is not None else src0.shape[0])
    rs = lambda t: t.stride(0) * t.element_size()
    with _device_guard(dst0, src0, dst1, src1):
        rc = _lib.lib().dga_copy_rows2(dst0.data_ptr() + dst0_off, rs(dst0), src0.data_ptr() + src0_off, rs(src0), bytes0,
                                       dst1.data_ptr() + dst1_off, rs(dst1), src1.data_ptr() + src1_off, rs(src1), bytes1,
                                       dst_index.data_ptr() if dst_index is not None else None,
                                       src_index.data_ptr() if src_index is not None else None, n, _stream_ptr(dst0))
        _lib.check(rc, "copy_rows2")


# ----------------------------------------------------------------------------- the framework's 16-bit entry points

def _dt16(t: torch.Tensor) -> int:
    if t.dtype == torch.bfloat16:
        return _lib.DT_BF16
    if t.dtype == torch.float16:
        return _lib.DT_FP16
    raise DGAError(-3, "dtype", f"expected bfloat16/float16, got {t.dtype}")


def run_mmad_rtc(x: torch.Tensor, y: torch.Tensor, z: torch.Tensor) -> None:
    """z[B,M,N] (f32, in place) = x[B,M,K] @ y[B,K,N]  (python_api.cpp:18, gemm.hpp:68-111). Synchronous, as the reference."""
    _require(x.dim() == 3 and y.dim() == 3 and z.dim() == 3, "rank must be 3")
    batch, m, k = x.shape
    _, k2, n = y.shape
    _require(k == k2 and y.shape[0] == batch and tuple(z.shape) == (batch, m, n), "shape mismatch")
    _require(z.dtype == torch.float32 and x.dtype == y.dtype, "dtype mismatch")
    for t in (x, y, z):
        _require(t.is_contiguous(), "operands must be contiguous")
    with _device_guard(x, y, z):
        ws_ptr, ws_bytes = _mmad_workspace(batch, m, n, k, x)
        rc = _lib.lib().dga_run_mmad_rtc_ws(x.data_ptr(), y.data_ptr(), z.data_ptr(), batch, m, n, k, _dt16(x),
                                            ws_ptr, ws_bytes, _stream_ptr(z))
        _lib.check(rc, "run_mmad_rtc")
        torch.cuda.current_stream(z.device).synchronize()  # gemm.hpp:110


def run_mmad_bench(x: torch.Tensor, y: torch.Tensor, z: torch.Tensor, params: torch.Tensor) -> None:
    """z[M,N] (f32) = x[M,K] @ y[K,N]; params int32[28]: slots 0..5 knobs in, 6..27 written back
    (python_api.cpp:23, gemm_bench.hpp:49-113)."""
    _require(x.dim() == 2 and y.dim() == 2 and z.dim() == 2, "rank must be 2")
    m, k = x.shape
    k2, n = y.shape
    _require(k == k2 and tuple(z.shape) == (m, n), "shape mismatch")
    _require(params.dtype == torch.int32 and params.numel() == 28, "params must be int32[28]")
    host = params.detach().cpu().tolist()          # the reference does 6 .item() syncs (gemm_bench.hpp:52-57)
    filled = bench_params_fill(m, n, k, host[:6])
    params.copy_(torch.tensor(filled, dtype=torch.int32))  # gemm_bench.hpp:79-81
    with _device_guard(x, y, z):
        arr = (ctypes.c_int32 * 28)(*filled)
        ws_ptr, ws_bytes = _mmad_workspace(1, m, n, k, x)
        rc = _lib.lib().dga_run_mmad_bench_ws(x.data_ptr(), y.data_ptr(), z.data_ptr(), m, n, k, _dt16(x), arr,
                                              ws_ptr, ws_bytes, _stream_ptr(z))
        _lib.check(rc, "run_mmad_bench")
        torch.cuda.current_stream(z.device).synchronize()


def run_mmad_custom(x: torch.Tensor, y: torch.Tensor, z: torch.Tensor) -> None:
    """The reference's static kernel returns immediately (include/impls/mmad.cpp:79): a no-op, kept for API parity."""
    return None
