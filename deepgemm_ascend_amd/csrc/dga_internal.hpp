// Internal (non-ABI) declarations shared by the host translation units.
#pragma once
#include <hip/hip_runtime_api.h>

#include "dga_hip.h"
#include <cstdint>

namespace dga {
int record_hip(hipError_t e);
// CUs of the current device (cached); the MI355X constant 256 when no device is visible (host-only tiling calls)
uint32_t device_cus();
// $DGA_DEFAULT_POLICY, read once: the arithmetic of a GEMM call that names no tiling.  1 = bf16-exact (the default: inside the
// operator's 2-ULP contract), 0 = "fast" (fp8 matrix instruction), 2 = "strict".
int default_policy();
// indexed masked-grouped form: flat row buffers + slot -> row table (dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed)
struct Fp8Indexed {
    const int64_t *row_index;   // device int64[groups * m_max]
    int64_t lda;                // bytes between rows of the flat A source
    int64_t sfa_ld;             // floats between rows of the scale source
    int64_t ldc;                // bf16 elements between rows of the flat destination
    int64_t rows;               // rows of the flat A source (bounds the buffer descriptor)
};
// row-strided dense form (dga_gemm_fp8_fp8_bf16_nt_strided): bytes between the rows of A and of B, DGA_ROWS_* flags
struct Fp8Strided {
    int64_t lda, ldb;
    int flags;
};
// the fp8 launcher behind the C-ABI GEMM entry points (dga_launch.hip); clock_stamps: see dga_diag.hip
int run_fp8(const void *a, const float *sfa, const void *b, const float *sfb, void *out, const int32_t *masked_m,
            const int32_t *m_indices, int b_groups, int groups, int m, int n, int k, int expected_m,
            const dga_tiling_t *tiling, void *workspace, size_t workspace_bytes, hipStream_t stream,
            unsigned long long *clock_stamps, const Fp8Indexed *ix, const Fp8Strided *sd = nullptr);
// compiled fp8 kernel menu (dga_launch.hip)
int variant_count();
void variant_info(int i, int *bm, int *bn, int *wm, int *wn, int *lds);
int variant_stages(int i);
bool variant_has_loader_waves(int i);   // a dispatchPolicyTag-4 build of that menu entry exists
// a 3-stage tiling on the plain loop whose tile has a loader-wave build takes it (same bits, 9-17 % less time)
void prefer_loader_waves(dga_tiling_t &t, bool upgrade_plain = true);
// odd-K re-layout pass (dga_rows.hip): rows of src_row_bytes at any byte alignment -> 16-byte aligned rows of dst_row_bytes, zero tail;
// two operands in one launch (rows1 == 0: one)
int pad_rows(const void *src0, void *dst0, int64_t rows0, const void *src1, void *dst1, int64_t rows1, int64_t src_row_bytes,
             int64_t dst_row_bytes, hipStream_t stream);
// the same with the sources' own row strides (bytes between rows, >= src_row_bytes)
int pad_rows_strided(const void *src0, int64_t stride0, void *dst0, int64_t rows0, const void *src1, int64_t stride1, void *dst1,
                     int64_t rows1, int64_t src_row_bytes, int64_t dst_row_bytes, hipStream_t stream);
// dense 256x256 tilings: turn a small last partial wave into a K-split tail (dga_tiling.cpp)
void apply_tail_split(dga_tiling_t &t, uint32_t cus);
}  // namespace dga
