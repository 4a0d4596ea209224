"""File-based golden / verify tools and the tuning sweep: the counterparts of the reference's
deep_gemm_ascend/scripts/{gen_golden,verify}.py and framework/benchmark/benchmark.py."""
