"""deepgemm_ascend_amd -- MI355X-native drop-in for the DeepGEMM_Ascend hot path.

Same entry-point names as the reference's ``deep_gemm_ascend`` package
(/root/reference/deep_gemm_ascend/framework/deep_gemm_ascend/__init__.py:1-4) plus the fp8
block-scaled operators.  Everything executes in libdga_hip.so (hand-written HIP for gfx950);
importing the operators without that library raises ``DGALibraryError``.
"""
from ._lib import DGAError, DGALibraryError, Platform, Problem, Tiling, build, lib  # noqa: F401
from .api import (  # noqa: F401
    CONFIG_FIELDS,
    bbit_params,
    bench_params_fill,
    catlass_dynamic_matmul,
    copy_rows,
    copy_rows2,
    gemm_fp8_fp8_bf16_nt,
    gemm_fp8_loop_clock,
    get_bench_config,
    get_best_config,
    infer_dtype,
    infer_shape,
    get_m_alignment_for_contiguous_layout,
    m_grouped_gemm_fp8_fp8_bf16_nt_contiguous,
    m_grouped_gemm_fp8_fp8_bf16_nt_masked,
    m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed,
    mfma_ceiling,
    per_block_cast_to_fp8,
    per_token_cast_to_fp8,
    platform_ascend910b,
    predict_time_us,
    route_slots,
    route_tokens,
    predictor_load,
    predictor_loaded,
    predictor_unload,
    release_scratch,
    select_kernel_with_predictor,
    select_tiling_strategy,
    platform_mi355x,
    run_mmad_bench,
    run_mmad_custom,
    run_mmad_rtc,
    select_kernel,
    tiling,
    tiling_check,
    tiling_cache_clear,
    tiling_cache_open,
    tiling_cache_size,
    workspace_bytes,
)

__all__ = [n for n in dir() if not n.startswith("_")]
