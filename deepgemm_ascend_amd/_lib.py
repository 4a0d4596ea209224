"""ctypes loader for libdga_hip.so (the C ABI declared in include/dga_hip.h).

There is no fallback: if the HIP library is missing or a symbol is absent the
import of the operator API fails loudly (``DGALibraryError``)."""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p
from pathlib import Path

_PKG = Path(__file__).resolve().parent
LIB_PATH = _PKG / "libdga_hip.so"


class DGALibraryError(RuntimeError):
    pass


class DGAError(RuntimeError):
    """Raised when a C-ABI call returns a negative status (the reference throws
    DGAException from DGA_HOST_ASSERT, csrc/utils/exception.hpp:9-33)."""

    def __init__(self, status: int, where: str, detail: str = ""):
        self.status = status
        msg = f"{where}: status {status} ({status_string(status)})"
        if detail:
            msg += f" {detail}"
        super().__init__(msg)


class Platform(ctypes.Structure):
    _fields_ = [("coreNum", c_uint32), ("ubSize", c_uint64), ("l1Size", c_uint64), ("l0ASize", c_uint64),
                ("l0BSize", c_uint64), ("l0CSize", c_uint64), ("xcdNum", c_uint32), ("waveSize", c_uint32)]


class Tiling(ctypes.Structure):
    _fields_ = [("strideA", c_uint64), ("strideB", c_uint64), ("strideC", c_uint64),
                ("m", c_uint32), ("n", c_uint32), ("k", c_uint32),
                ("m1", c_uint16), ("n1", c_uint16), ("k1", c_uint16),
                ("swizzleOffset", c_uint8), ("swizzleDirection", c_uint8), ("splitkFactor", c_uint16),
                ("layoutTagA", c_uint8), ("layoutTagB", c_uint8), ("layoutTagC", c_uint8),
                ("paddingTagA", c_uint8), ("paddingTagB", c_uint8), ("paddingTagC", c_uint8),
                ("kernelSerial", c_uint8), ("dispatchPolicyTag", c_uint8), ("build", c_uint8), ("reserved0", c_uint8),
                ("blockDim", c_uint32),
                ("wavesM", c_uint8), ("wavesN", c_uint8), ("stages", c_uint8), ("contiguous", c_uint8),
                ("ldsBytes", c_uint32), ("groups", c_uint32)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


class Problem(ctypes.Structure):
    _fields_ = [("m", c_uint32), ("n", c_uint32), ("k", c_uint32), ("groups", c_uint32), ("expected_m", c_uint32),
                ("layoutTagA", c_uint8), ("layoutTagB", c_uint8), ("layoutTagC", c_uint8), ("dtype", c_uint8),
                ("flags", c_uint32)]


PROBLEM_CONTIGUOUS_M = 1
# dga_tiling_t.build (include/dga_hip.h DGA_BUILD_*)
BUILD_DEFAULT, BUILD_WSK_REGISTER, BUILD_BX_AIMAGE, BUILD_BX_IMAGE8, BUILD_BX_IMAGE4 = 0, 1, 4, 5, 6
BUILD_BX_PERSISTENT, BUILD_BX_ONE_TILE, BUILD_BX_GROUPED, BUILD_BX_DECODE = 7, 8, 9, 10
ROWS_A_ZERO_PADDED, ROWS_B_ZERO_PADDED = 1, 2   # dga_gemm_fp8_fp8_bf16_nt_strided flags
CAST_UE8M0 = 1                                  # dga_cast_to_fp8_*_ex flag: block scales rounded up to powers of two
CONTIGUOUS_M_ALIGNMENT = 128


class ShardedShape(ctypes.Structure):      # dga_sharded_shape_t
    _fields_ = [("world", c_int32), ("rank", c_int32), ("groups_total", c_int32), ("m_max", c_int32), ("n", c_int32),
                ("k", c_int32), ("chunks", c_int32), ("max_tokens", c_int32), ("capacity_factor", c_float),
                ("indexed", c_int32), ("policy", c_int32)]


class ShardedLayout(ctypes.Structure):     # dga_sharded_layout_t
    _fields_ = [("groups_local", c_int32), ("groups_per_chunk", c_int32), ("chunks", c_int32), ("kb", c_int32), ("nb", c_int32),
                ("indexed", c_int32), ("hdr_offset", c_int64), ("row_bytes", c_int64), ("pair_capacity", c_int64),
                ("rows_per_chunk", c_int64), ("rows_total", c_int64), ("max_tokens", c_int64),
                ("send_bytes", ctypes.c_uint64), ("recv_bytes", ctypes.c_uint64), ("osend_bytes", ctypes.c_uint64),
                ("oback_bytes", ctypes.c_uint64), ("slot_bytes", ctypes.c_uint64), ("rdest_bytes", ctypes.c_uint64),
                ("row_of_slot_bytes", ctypes.c_uint64), ("pair_cnt_bytes", ctypes.c_uint64), ("masked_m_bytes", ctypes.c_uint64),
                ("packed_a_bytes", ctypes.c_uint64), ("packed_sfa_bytes", ctypes.c_uint64), ("packed_out_bytes", ctypes.c_uint64),
                ("events", c_int32), ("steps", c_int32)]


class ShardedBuffers(ctypes.Structure):    # dga_sharded_buffers_t
    _fields_ = [("send", c_void_p), ("recv", c_void_p), ("osend", c_void_p), ("oback", c_void_p), ("slot", c_void_p),
                ("rdest", c_void_p), ("row_of_slot", c_void_p), ("pair_cnt", c_void_p), ("masked_m", c_void_p),
                ("overflow", c_void_p), ("packed_a", c_void_p), ("packed_sfa", c_void_p), ("packed_out", c_void_p),
                ("b", c_void_p), ("sfb", c_void_p), ("workspace", c_void_p), ("workspace_bytes", c_size_t)]


class ShardedStep(ctypes.Structure):       # dga_sharded_step_t
    _fields_ = [("op", c_int32), ("stream", c_int32), ("chunk", c_int32), ("event", c_int32), ("row_begin", c_int64),
                ("rows", c_int64), ("group_begin", c_int32), ("groups", c_int32)]


(STEP_WAIT_EVENT, STEP_RECORD_EVENT, STEP_CLEAR_HEADERS, STEP_ROUTE_SOURCE, STEP_PACK, STEP_ZERO_COUNTS, STEP_ZERO_DROPPED,
 STEP_ALL_TO_ALL_DISPATCH, STEP_ROUTE_RECEIVED, STEP_UNPACK, STEP_GEMM, STEP_GATHER_OUT, STEP_ALL_TO_ALL_COMBINE,
 STEP_RESTORE_ORDER, STEP_ZERO_UNROUTED) = range(15)
ALL_TO_ALL_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p)


DT_FP16, DT_BF16, DT_FP8_E4M3FN, DT_FP32 = 1, 2, 3, 4
LAYOUT_ROW_MAJOR, LAYOUT_COLUMN_MAJOR = 0, 1

# name -> (restype, argtypes): every symbol include/dga_hip.h declares
SIGNATURES = {
    "dga_infer_shape": (c_int, [POINTER(c_int64), c_int, POINTER(c_int64), c_int, POINTER(c_int64)]),
    "dga_infer_dtype": (c_int, [c_int, c_int, POINTER(c_int)]),
    "dga_tiling": (c_int, [POINTER(Problem), POINTER(Tiling)]),
    "dga_tiling_bf16_exact": (c_int, [POINTER(Problem), POINTER(Tiling)]),
    "dga_select_kernel": (c_int, [POINTER(Problem), POINTER(Platform), POINTER(Tiling)]),
    "dga_platform_mi355x": (None, [POINTER(Platform)]),
    "dga_platform_ascend910b": (None, [POINTER(Platform), c_uint32]),
    "dga_predictor_load": (c_int, [c_char_p]),
    "dga_predictor_unload": (None, []),
    "dga_predictor_loaded": (c_int, []),
    "dga_predict_time_us": (c_int, [POINTER(Problem), POINTER(Tiling), POINTER(ctypes.c_float)]),
    "dga_select_tiling_strategy": (c_int, [POINTER(ctypes.c_float), POINTER(ctypes.c_int32), c_int, c_int, c_int, ctypes.c_float, c_int,
                                           ctypes.c_uint64, POINTER(c_int), POINTER(c_int), POINTER(c_int)]),
    "dga_select_kernel_with_predictor_ex": (c_int, [POINTER(Problem), POINTER(Tiling), POINTER(ctypes.c_float), POINTER(ctypes.c_float),
                                                    c_int, c_int]),
    "dga_select_kernel_with_predictor": (c_int, [POINTER(Problem), POINTER(Tiling), POINTER(ctypes.c_float),
                                                 POINTER(ctypes.c_float)]),
    "dga_tiling_cache_open": (c_int, [c_char_p]),
    "dga_tiling_cache_clear": (c_int, []),
    "dga_tiling_cache_size": (c_int, []),
    "dga_workspace_bytes": (c_size_t, [POINTER(Tiling)]),
    "dga_gemm_fp8_fp8_bf16_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         POINTER(Tiling), c_void_p, c_size_t, c_void_p]),
    "dga_gemm_fp8_fp8_bf16_nt_strided": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int,
                                                 c_int, c_int, POINTER(Tiling), c_void_p, c_size_t, c_void_p]),
    "dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                          c_int, c_int, c_int, c_int, c_int, POINTER(Tiling),
                                                          c_void_p, c_size_t, c_void_p]),
    "dga_m_grouped_gemm_fp8_fp8_bf16_nt_contiguous": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                                              c_void_p, c_int, c_int, c_int, c_int, POINTER(Tiling),
                                                              c_void_p, c_size_t, c_void_p]),
    "dga_cast_to_fp8_1x128": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "dga_cast_to_fp8_128x128": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]),
    "dga_cast_to_fp8_1x128_ld": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "dga_cast_to_fp8_128x128_ld": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "dga_cast_to_fp8_1x128_ex": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "dga_cast_to_fp8_128x128_ex": (c_int, [c_void_p, c_int, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int, c_void_p]),
    "dga_catlass_dynamic_matmul_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_void_p, c_void_p]),
    "dga_catlass_dynamic_matmul": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                           c_void_p]),
    "dga_get_best_config": (c_int, [c_uint32] * 4 + [POINTER(c_uint32)]),
    "dga_get_bench_config": (c_int, [c_uint32] * 9 + [POINTER(c_uint32)]),
    "dga_bench_params_fill": (c_int, [c_uint32] * 3 + [POINTER(c_int32)]),
    "dga_bbit_params": (c_int, [c_uint32] * 9 + [POINTER(c_uint32)]),
    "dga_run_mmad_rtc": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dga_run_mmad_bench": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int32),
                                   c_void_p]),
    "dga_mmad_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int, c_void_p]),
    "dga_run_mmad_rtc_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_size_t,
                                    c_void_p]),
    "dga_run_mmad_bench_ws": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, POINTER(c_int32),
                                      c_void_p, c_size_t, c_void_p]),
    "dga_route_tokens": (c_int, [c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "dga_route_slots": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    "dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p,
                                                                  c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_int,
                                                                  c_int, c_int, c_int, c_int, POINTER(Tiling), c_void_p,
                                                                  c_size_t, c_void_p]),
    "dga_copy_rows2": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int64, c_int64,
                               c_void_p, c_void_p, c_int64, c_void_p]),
    "dga_copy_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int64, c_void_p]),
    "dga_gemm_fp8_loop_clock": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, POINTER(Tiling),
                                        c_void_p, c_size_t, c_int, c_void_p, POINTER(c_float), POINTER(c_float)]),
    "dga_sharded_layout": (c_int, [POINTER(ShardedShape), POINTER(ShardedLayout)]),
    "dga_sharded_plan": (c_int, [POINTER(ShardedShape), POINTER(ShardedStep), c_int, POINTER(c_int)]),
    "dga_sharded_forward": (c_int, [POINTER(ShardedShape), POINTER(ShardedBuffers), c_void_p, c_void_p, c_void_p, c_int, c_void_p,
                                    c_int, POINTER(c_void_p), POINTER(c_void_p), ALL_TO_ALL_FN, c_void_p]),
    "dga_sharded_events_create": (c_int, [c_int, POINTER(c_void_p)]),
    "dga_sharded_events_destroy": (c_int, [c_int, POINTER(c_void_p)]),
    "dga_mfma_ceiling": (c_int, [c_int, c_int, c_void_p, c_size_t, c_void_p, POINTER(c_float)]),
    "dga_tiling_check": (c_int, [POINTER(Tiling)]),
    "dga_default_policy": (c_int, [ctypes.c_char_p, c_int]),
    "dga_status_string": (c_char_p, [c_int]),
    "dga_last_hip_error": (c_int, []),
    "dga_abi_version": (c_int, []),
    "dga_device_platform": (c_int, [POINTER(Platform)]),
}

_lib = None


def build(force: bool = False) -> Path:
    """Compile libdga_hip.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    srcs = list((_PKG / "csrc").glob("*")) + [(_PKG.parent / "include" / "dga_hip.h")]
    newest = max(p.stat().st_mtime for p in srcs if p.is_file())
    if force or not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < newest:
        subprocess.check_call(["make", "-C", str(_PKG / "csrc"), "all"])
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise DGALibraryError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C {_PKG / 'csrc'}`. There is no CPU fallback.")
        try:
            L = ctypes.CDLL(str(LIB_PATH))
        except OSError as e:  # e.g. libamdhip64 not found
            raise DGALibraryError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(L, name)
            except AttributeError as e:
                raise DGALibraryError(f"{LIB_PATH} does not export {name}") from e
            fn.restype = res
            fn.argtypes = args
        if L.dga_abi_version() != 7:
            raise DGALibraryError("ABI version mismatch")
        _lib = L
    return _lib


def status_string(status: int) -> str:
    try:
        return lib().dga_status_string(status).decode()
    except Exception:  # pragma: no cover
        return "?"


def check(status: int, where: str):
    if status != 0:
        detail = ""
        if status == -5:
            detail = f"(hipError {lib().dga_last_hip_error()})"
        raise DGAError(status, where, detail)
