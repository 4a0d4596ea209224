/*
 * dga_hip.h -- C ABI of libdga_hip.so: the MI355X (gfx950) drop-in for the
 * DeepGEMM_Ascend hot path.  Plain pointers and sizes only; every entry point
 * returns an int status (0 = DGA_OK, negative = error) and never throws.
 * Device pointers are caller-owned; nothing here allocates device memory.
 * `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to /root/reference).
 */
#ifndef DGA_HIP_H
#define DGA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGA_ABI_VERSION 7

/* ---- status codes (reference: DGA_HOST_ASSERT throws DGAException,
 *      deep_gemm_ascend/framework/csrc/utils/exception.hpp:9-33; op hooks return
 *      ge::GRAPH_FAILED, aclnn_catlass_dynamic_matmul/op_host/catlass_dynamic_matmul_tiling.cpp:86-100) */
enum {
    DGA_OK = 0,
    DGA_E_NULL = -1,      /* required pointer is NULL */
    DGA_E_SHAPE = -2,     /* rank / dimension mismatch (InferShape / TilingFunc checks) */
    DGA_E_DTYPE = -3,     /* dtype mismatch (InferDataType) */
    DGA_E_ALIGN = -4,     /* pointer / stride alignment the kernel cannot take */
    DGA_E_HIP = -5,       /* a HIP runtime call failed (see dga_last_hip_error) */
    DGA_E_TILING = -6,    /* tiling not supported by any compiled kernel variant */
    DGA_E_WORKSPACE = -7, /* workspace too small */
    DGA_E_IO = -8,        /* harness / cache file error */
    DGA_E_RANGE = -9      /* masked_m / knob out of range */
};

/* dtypes (the values the aclnn op proto uses are ge::DataType; we keep our own small enum) */
enum { DGA_DT_FP16 = 1, DGA_DT_BF16 = 2, DGA_DT_FP8_E4M3FN = 3, DGA_DT_FP32 = 4 };

/* LayoutTag / PaddingTag: aclnn_catlass_dynamic_matmul/op_host/op_tiling/tiling_params.h:16-17 */
enum { DGA_LAYOUT_ROW_MAJOR = 0, DGA_LAYOUT_COLUMN_MAJOR = 1 };
enum { DGA_PADDING_NONE = 0, DGA_PADDING_ND = 1, DGA_PADDING_BLOCK_ND = 2, DGA_PADDING_NZ = 3 };

/* kernelSerial menu, same numbering as the reference
 * (op_kernel/kernel/kernel_utils.h:31-37, select_kernel.cpp:270-331):
 *   0 Common, 1 Small, 2 PaddingCommon (K % 16 != 0 read in place by the loader waves of the 128 x 256 tile: the re-layout fused
 *   with the matmul as in the reference's kernel of that name; without it odd K takes a padding pass), 4 StreamK/split-K. */
enum { DGA_KERNEL_COMMON = 0, DGA_KERNEL_SMALL = 1, DGA_KERNEL_PADDING_COMMON = 2, DGA_KERNEL_STREAMK = 4,
       DGA_KERNEL_STREAMK_TAIL = 5 /* whole waves of 256x256 tiles, the last partial wave covered by 128x128 tiles in a second launch;
                                      under DGA_POLICY_BF16_EXACT: whole waves of 128x256 tiles + 64x128 quarter tiles.  Same bytes as
                                      the single launch; a tail longer than half the CUs runs as the single launch */,
       DGA_KERNEL_SPLITK_WORKGROUP = 6 /* M <= 64: the 8 waves of a workgroup are the 8 K slices of one output tile, partial tiles
                                          combined in LDS -- one launch, no slab (the reference's single-core split-K kernel types,
                                          op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36); the bits of split-K with factor 8.
                                          M <= 32: operands staged through per-wave LDS-DMA rings (tiling.stages = 1 names the
                                          older build that streams fragments global -> registers instead) */,
       DGA_KERNEL_STREAMK_ONE_LAUNCH = 7 /* Stream-K proper (the reference's kernel type 4, padding_streamk_matmul_kernel.h:94-98): ONE
                                          launch of one workgroup per CU; the raster's (tile, k block) units are cut evenly over the
                                          workgroups, a run that ends inside a tile leaves an fp32 partial tile in the workspace and the
                                          workgroup holding the tile's first k blocks adds the partials in k order.  256 x 256 tile, dense,
                                          M and N multiples of 256, K of 128; anything else runs the tiling's tile kernel.  Needs
                                          dga_workspace_bytes() of workspace (256 KB per CU) */ };

/* dispatchPolicyTag of the fp8 tile kernels (the reference's field selects a catlass dispatch policy,
 * op_tiling/tiling_params.h:19-66; here it selects the main-loop schedule or the exact-arithmetic kernel):
 *   0 plain (one barrier per k block), 1 ping-pong, 2 continuous pipeline -- the same results, bit for bit;
 *   3 strict: fp32-input MFMA chain in the reference CPU path's own order (fp32 products, fp32 running sum, k ascending:
 *     framework/tests/test.py:37) -- bit-identical to the oracle, any shape, at the fp32 matrix rate (1/32 of the fp8
 *     rate).  $DGA_STRICT=1 forces it for every fp8 call of the process.
 *   4 loader waves: the plain loop with four extra waves that only issue the LDS-DMA (128x256 tile, 3 stages): the masked
 *     grouped weight stream (-2 % time) and dense problems of about one such tile per CU (BASELINE configs[2]: -9 %);
 *     same bits as policy 0; a tile without such a build runs policy 0.
 *   5 persistent loader waves: policy 4 with one workgroup per CU that walks its share of the tiles, the LDS ring running
 *     across tile boundaries (the next tile's first k blocks are in flight while this one is stored) -- what the
 *     reference's one-block-per-AI-core kernel is by construction (framework/csrc/jit/generate_code.hpp:160-198); same
 *     bits as policy 0; whole rasters only (split-K and the quarter-tile tail run policy 4).
 *   6 persistent continuous pipeline: policy 2 (256x256 tile) with one workgroup per CU walking its tiles, the refill slots
 *     of a tile's last two k blocks fetching the next tile's first two; dense rasters of full tiles (M, N multiples of 256,
 *     K of 128, at least two k blocks) -- anything else runs policy 2; same bits.
 *   7 bf16-exact: the e4m3 bytes are up-converted to bf16 in registers (exact) and every 128-wide scale block is summed by four
 *     chained v_mfma_f32_16x16x32_bf16 -- exact products, fp32-class sums (2^-25 of the block's sum of magnitudes, against
 *     2^-16 on the fp8 matrix instruction), the same fp32 promotion.  The policy between the fast path (policies 0-2, 4-6) and
 *     the strict one: within 2 bf16 ULP of the reference CPU path (framework/tests/test.py:19-64) on all but ~1e-6 of the
 *     outputs of BASELINE configs[1] (every one a sum that cancels to < 2^-21 of its terms), at the bf16 matrix rate.  Takes
 *     every layout the tile kernels take (dense, masked, contiguous, indexed, split-K); the tile comes from the tiling's
 *     (m1, n1) mapped onto the policy's own menu (wave tiles of at most 64 x 64).  $DGA_BF16_EXACT=1 forces it for every fp8
 *     call of the process.
 *   | 16 (DGA_POLICY_UE8M0_SCALES, a FLAG on the fast-path schedules 0, 2, 4, 5, 6): the caller's promise that every value of sfa
 *     and sfb is an exact power of two in the normal fp32 range ("UE8M0" scales: 2^ceil(log2(amax / 448)), what upstream DeepGEMM's
 *     per_token_cast_to_fp8(..., use_ue8m0=True) writes).  The scales then ride in the E8M0 operands of
 *     v_mfma_scale_f32_16x16x128_f8f6f4 and the MFMA accumulates in place, as the reference's Mmad(c1Local, ..., init on first)
 *     does (framework/csrc/jit/generate_code.hpp:320-335): no promotion on the vector pipe.  Outputs equal the promotion form's up
 *     to fp32 rounding order (profiles/r05_probe_scale_acc.txt: identical bf16 on 76 800 of 76 800 outputs); a tile without such a
 *     build runs the promotion form.  A scale that is NOT a power of two is read as its exponent alone (mantissa dropped); zero
 *     reads as 2^-127.  Ignored by policy 3.
 *     On policy 7 (7 | 16): the bf16-exact arithmetic with the scales folded into the e4m3 -> bf16 conversions of the A fragments (a
 *     power of two times an e4m3 value is a bf16 value: exact) -- the bf16 MFMA chain then accumulates through every k block in
 *     place, no promotion; 128 x 256 and 64 x 256 tiles, every layout; other tiles run policy 7 as it is. */
enum { DGA_POLICY_PLAIN = 0, DGA_POLICY_PINGPONG = 1, DGA_POLICY_CONTINUOUS = 2, DGA_POLICY_STRICT = 3,
       DGA_POLICY_LOADER_WAVES = 4, DGA_POLICY_PERSISTENT = 5, DGA_POLICY_CONTINUOUS_PERSISTENT = 6,
       DGA_POLICY_BF16_EXACT = 7, DGA_POLICY_UE8M0_SCALES = 16 };

/* dga_tiling_t.build: the compiled build a caller (a sweep, a test, a tuned CSV row) names beside the tile and the schedule.  0 lets
 * the dispatcher choose (what every selector of this library writes unless it says otherwise below).  No reference counterpart: the
 * reference compiles one kernel per tiling key (op_kernel/catlass_dynamic_matmul_tiling_key.h:30-36).
 *   fast path     DGA_BUILD_WSK_REGISTER with kernelSerial 6 only: fragments global -> registers (M <= 64) instead of the LDS-DMA rings
 *   bf16-exact    DGA_BUILD_BX_AIMAGE / _IMAGE8 / _IMAGE4: the 128 x 256 tile with A (8 waves) or both operands (8 / 4 waves) converted
 *                 once per workgroup into a bf16 LDS image;  DGA_BUILD_BX_PERSISTENT / _ONE_TILE: one workgroup per CU walking the raster /
 *                 one workgroup per tile (0: persistent on rasters of more than one round);  DGA_BUILD_BX_GROUPED: the masked grouped
 *                 layout's kernel (two k blocks of the ring in flight, the loop unrolled for the m-tiles that hold rows) -- what
 *                 dga_tiling_bf16_exact names for masked grouped problems;  DGA_BUILD_BX_DECODE with kernelSerial 6 and the 64 x 128 tile
 *                 only: the one-launch split-K for a few 64-row tiles (two k groups per workgroup, splitkFactor <= 8 workgroups per tile
 *                 whose fp32 partial tiles meet in the workspace -- dga_workspace_bytes counts them; a launch it does not take, e.g. no
 *                 workspace or more tiles than CUs, runs the two-launch split-K of the same tiling);  DGA_BUILD_WSK_REGISTER as above.
 * The values are the magic `stages` values of ABI <= 6, so a CSV row written then (stages 1, 4..8) maps onto (build = stages,
 * stages = 3) when it is read. */
enum { DGA_BUILD_DEFAULT = 0, DGA_BUILD_WSK_REGISTER = 1, DGA_BUILD_BX_AIMAGE = 4, DGA_BUILD_BX_IMAGE8 = 5, DGA_BUILD_BX_IMAGE4 = 6,
       DGA_BUILD_BX_PERSISTENT = 7, DGA_BUILD_BX_ONE_TILE = 8, DGA_BUILD_BX_GROUPED = 9, DGA_BUILD_BX_DECODE = 10 };

/* Platform description: the CDNA4 retarget of PlatformInfo
 * (op_tiling/platform_info.h:16-41; Python mirror get_best_config/tiling_calculator.py:25-30).
 * The same struct carries the Ascend numbers when the reference's own arithmetic is
 * replayed for parity (tests only). */
typedef struct dga_platform_t {
    uint32_t coreNum;   /* Ascend: AI cores (24/20).  MI355X: CUs (256) */
    uint64_t ubSize;    /* Ascend UB 192 KiB.       MI355X: unused (0) */
    uint64_t l1Size;    /* Ascend L1 512 KiB.       MI355X: LDS per CU (160 KiB) */
    uint64_t l0ASize;   /* Ascend 64 KiB.           MI355X: VGPR bytes per SIMD lane-set usable for A fragments */
    uint64_t l0BSize;
    uint64_t l0CSize;   /* Ascend L0C 128 KiB.      MI355X: fp32 accumulator bytes per workgroup (8 waves x 128 regs x 64 lanes x 4) */
    uint32_t xcdNum;    /* MI355X: 8 (Ascend: 1) */
    uint32_t waveSize;  /* MI355X: 64 */
} dga_platform_t;

/* Host-side superset of the kernel tiling data: TilingParams
 * (op_tiling/tiling_params.h:19-66) + CatlassDynamicMatmulTilingData
 * (op_kernel/catlass_dynamic_matmul_tiling_data.h:19-33) + CDNA4 fields. */
typedef struct dga_tiling_t {
    uint64_t strideA, strideB, strideC;
    uint32_t m, n, k;
    uint16_t m1, n1, k1;         /* workgroup tile BM x BN x BK */
    uint8_t swizzleOffset;       /* tile-rows rastered together (tiling_params.h:63) */
    uint8_t swizzleDirection;    /* (m > n) ? 0 : 1 (tiling_params.h:64) */
    uint16_t splitkFactor;
    uint8_t layoutTagA, layoutTagB, layoutTagC;
    uint8_t paddingTagA, paddingTagB, paddingTagC;
    uint8_t kernelSerial;
    uint8_t dispatchPolicyTag;
    uint8_t build;               /* DGA_BUILD_*: which compiled build of (kernelSerial, dispatchPolicyTag, tile) runs; 0 = the
                                    dispatcher's rule.  ABI 7: until ABI 6 these names rode on magic values of `stages` */
    uint8_t reserved0;           /* 0 */
    uint32_t blockDim;           /* workgroups launched (reference: uint8 AI-core count) */
    /* CDNA4 */
    uint8_t wavesM, wavesN;      /* wave grid inside the workgroup */
    uint8_t stages;              /* LDS stages, as in the reference's tiling data (op_kernel/catlass_dynamic_matmul_tiling_data.h:19-33
                                    has none: the catlass dispatch policy fixes them): 0 = the tile's default, 2 or 3 */
    uint8_t contiguous;          /* 1 = contiguous-grouped layout (groups = number of B matrices) */
    uint32_t ldsBytes;
    uint32_t groups;             /* 1 = dense */
} dga_tiling_t;

typedef struct dga_problem_t {
    uint32_t m, n, k;
    uint32_t groups;             /* 1 = dense; >1 = grouped masked-M with m = m_max */
    uint32_t expected_m;         /* grouped: hint for tile choice (0 = m) */
    uint8_t layoutTagA, layoutTagB, layoutTagC;
    uint8_t dtype;               /* DGA_DT_* of the inputs */
    uint32_t flags;              /* DGA_PROBLEM_* */
} dga_problem_t;

#define DGA_PROBLEM_CONTIGUOUS_M 1u   /* contiguous-grouped layout: tile height <= DGA_CONTIGUOUS_M_ALIGNMENT */
#define DGA_CONTIGUOUS_M_ALIGNMENT 128

/* ---- operator hooks -------------------------------------------------------------------- */

/* InferShape: aclnn_catlass_dynamic_matmul/op_host/catlass_dynamic_matmul.cpp:16-35.
 * self [m,k], mat2 logical [k,n] -> out [m,n]; both ranks must be 2. */
int dga_infer_shape(const int64_t *self_shape, int self_rank, const int64_t *mat2_shape, int mat2_rank,
                    int64_t *out_shape /*[2]*/);

/* InferDataType: catlass_dynamic_matmul.cpp:37-46 (inputs must match; out = in).
 * fp8 inputs are the extension: out = bf16. */
int dga_infer_dtype(int self_dtype, int mat2_dtype, int *out_dtype);

/* TilingFunc: catlass_dynamic_matmul_tiling.cpp:77-122 = shape checks + TilingParams ctor +
 * SelectKernelWithCache (select_kernel.cpp:371-378).  Consults the (m,n,k)-keyed tiling cache,
 * optionally CSV-backed through $CACHE_FILE_PATH / $DGA_CACHE_FILE_PATH (cache.cpp:22-101). */
int dga_tiling(const dga_problem_t *problem, dga_tiling_t *out);

/* TilingFunc of the bf16-exact arithmetic policy (dispatchPolicyTag 7): dga_tiling, then -- for dense problems -- the tile and
 * split-K factor picked from that policy's own menu by its own cost model (wave tiles <= 64 x 64, one 8-wave build; the fast path's
 * tuned tile is 10-40 % off there on mid-M shapes).  out->dispatchPolicyTag = DGA_POLICY_BF16_EXACT.  No reference counterpart. */
int dga_tiling_bf16_exact(const dga_problem_t *problem, dga_tiling_t *out);

/* Does the compiled kernel menu hold this tiling?  DGA_OK, or the error every fp8 GEMM entry returns for it BEFORE any launch
 * (they call the same check first): DGA_E_TILING for a (kernelSerial, dispatchPolicyTag, m1 x n1, wavesM x wavesN, stages) no build
 * answers to, DGA_E_RANGE for a split factor beyond 1024.  A caller-written dga_tiling_t is data from outside; the counterpart of
 * CatlassDynamicMatmulTilingFunc returning GRAPH_FAILED (op_host/catlass_dynamic_matmul_tiling.cpp:86-100).  What the fields may
 * hold:  kernelSerial 0, 1, 2, 4, 5, 6, 7;  k1 0 or 128;  dispatchPolicyTag 0..7, optionally | DGA_POLICY_UE8M0_SCALES;  stages 0, 2 or 3;
 * reserved0 0;
 *   policy 3 (strict): any tile, stages and build (the kernel picks its own);
 *   policy 7 (bf16-exact): any m1, n1 > 0 (mapped onto that policy's menu); build 0 or a DGA_BUILD_BX_* name (DGA_BUILD_WSK_REGISTER only
 *     with kernelSerial 6; DGA_BUILD_BX_DECODE only with kernelSerial 6 on m1 x n1 = 64 x 128);
 *   fast path: m1 x n1 a tile of the menu, wavesM x wavesN either 0 x 0 or a wave grid that tile is built with; build 0
 *     (DGA_BUILD_WSK_REGISTER only with kernelSerial 6); policy 1 and kernelSerial 5 / 7: 256 x 256 only. */
int dga_tiling_check(const dga_tiling_t *tiling);

/* The arithmetic of an fp8 call that names neither a policy nor a tiling: $DGA_DEFAULT_POLICY, parsed and validated ONCE per process,
 * here, for every front end (the C entry points with tiling == NULL, deepgemm_ascend_amd/api.py, the deep_gemm_cpp extension).
 * Names: "bf16_exact" (the default: inside the operator's 2-ULP contract), "fast", "strict", "fast_ue8m0", "bf16_exact_ue8m0" (the
 * caller's promise of power-of-two scales for every call), "auto" (bf16-exact where the decode kernel carries it, fast elsewhere).
 * Writes the NUL-terminated name to name_out[cap] (nullable) and returns DGA_OK -- or DGA_E_RANGE for a value that is none of these
 * (an unset or empty variable is "bf16_exact"): a typo must not silently change the arithmetic; every fp8 entry that needs the default
 * returns the same error.  No reference counterpart (its one kernel has one arithmetic). */
int dga_default_policy(char *name_out, int cap);

/* SelectKernel without the cache, on an explicit platform (select_kernel.cpp:333-369).
 * platform == NULL -> MI355X.  With dga_platform_ascend910b() it replays the reference's
 * DoTilingLayout01 + handler chain for parity tests. */
int dga_select_kernel(const dga_problem_t *problem, const dga_platform_t *platform, dga_tiling_t *out);

void dga_platform_mi355x(dga_platform_t *out);
void dga_platform_ascend910b(dga_platform_t *out, uint32_t core_num /*24 C++ default, 20 Python default*/);

/* Learned predictor (get_best_config/get_best_config.py:166-670 TilingPredictor + model.py TimePredictMLP; C++ hook
 * op_tiling/predictor.cpp:107-157, SelectKernelWithPredictor select_kernel.cpp:380-388).  The weights file is the
 * text export of harness/train_predictor.py; tuned/predictor_mi355x.txt next to the library is loaded on first use
 * unless $DGA_NO_PREDICTOR is set.  dga_tiling() consults it on a cache miss for dense fp8 problems.
 *   dga_predictor_load(NULL) = the default file; DGA_E_IO if the file is missing or malformed.
 *   dga_select_kernel_with_predictor: native tiling (dga_select_kernel) unless the model's greedy pick over the
 *   compiled candidates promises >= 20 % over it (the reference: 3 %; raised by measurement against the fitted selector) and there are >= 4 candidates (the reference's two fallbacks,
 *   get_best_config.py:587-621); *predicted_us / *native_us = the model's times for the result / the native tiling
 *   (0 when no model is loaded or the problem is outside what the model covers). */
int dga_predictor_load(const char *path);
void dga_predictor_unload(void);
int dga_predictor_loaded(void);
int dga_predict_time_us(const dga_problem_t *problem, const dga_tiling_t *tiling, float *us);
/* Selection strategies of the predictor (get_best_config.py:431-525 select_tiling_strategy): greedy, topk_median, topk_dbscan.
 * preds[count] = predicted times, tiles[count][3] = (mTile, nTile, kTile) of every candidate (used by topk_dbscan only).
 * *picked_index = the candidate taken; topk_dbscan also returns the winning cluster (cluster_members[<= count], *cluster_count;
 * both nullable) -- the reference draws a random member of it (random.Random(random_state).choice), this library takes its fastest
 * member (random_state == 0) or member random_state % size, so that a pick is reproducible. */
enum { DGA_PICK_GREEDY = 0, DGA_PICK_TOPK_MEDIAN = 1, DGA_PICK_TOPK_DBSCAN = 2 };
int dga_select_tiling_strategy(const float *preds, const int32_t *tiles, int count, int method, int topk, float dbscan_eps,
                               int dbscan_min_samples, uint64_t random_state, int *picked_index, int *cluster_members, int *cluster_count);
/* dga_select_kernel_with_predictor with the strategy named (method: DGA_PICK_*, topk as in the reference: 10). */
int dga_select_kernel_with_predictor_ex(const dga_problem_t *problem, dga_tiling_t *out, float *predicted_us, float *native_us,
                                        int method, int topk);
int dga_select_kernel_with_predictor(const dga_problem_t *problem, dga_tiling_t *out, float *predicted_us,
                                     float *native_us);

/* Tiling cache control (TilingCache, cache.cpp:69-100; CSV::Document, csv.cpp:31-140). */
int dga_tiling_cache_open(const char *csv_path);  /* NULL/"" = memory only */
int dga_tiling_cache_clear(void);
int dga_tiling_cache_size(void);

/* workspace: the reference asks the runtime for a fixed 200 MB
 * (catlass_dynamic_matmul_tiling.cpp:115-120); we size it from the tiling (0 unless split-K). */
size_t dga_workspace_bytes(const dga_tiling_t *tiling);

/* ---- the hot path ---------------------------------------------------------------------- */

/* gemm_fp8_fp8_bf16_nt: out[M,N] (bf16) = dequant(A[M,K] e4m3fn, sfa[M,ceil(K/128)] f32)
 *                                       . dequant(B[N,K] e4m3fn, sfb[ceil(N/128),ceil(K/128)] f32)^T
 * Replaces the launch half of run_mmad_rtc / mmad_rtc
 * (deep_gemm_ascend/framework/csrc/python_api.cpp:18, jit_kernels/impls/gemm.hpp:68-111) and of the
 * aclnn op's device entry (op_kernel/catlass_dynamic_matmul.cpp:16-45), NT layout as in
 * catlass_dynamic_matmul_tiling.cpp:83-84.  All rows contiguous (lda = ldb = K, ldc = N).
 * tiling == NULL -> dga_tiling() is called.  Asynchronous on `stream`. */
int dga_gemm_fp8_fp8_bf16_nt(const void *a, const float *sfa, const void *b, const float *sfb, void *out,
                             int m, int n, int k, const dga_tiling_t *tiling, void *workspace,
                             size_t workspace_bytes, void *stream);

/* The same with the operands' own row strides (bytes between rows: lda, ldb >= K, each either K or a multiple of 16; out and the
 * scales stay contiguous) -- what a framework hands over when its tensors are views of padded buffers (the reference's Python
 * entry takes torch tensors, strides and all: deep_gemm_ascend/framework/csrc/python_api.cpp:18).  An operand whose rows start
 * on 16-byte boundaries is read in place.  For K % 16 != 0 that needs the bytes from K to the next 16-byte boundary of every row to
 * be zero, which the caller states per operand with DGA_ROWS_A_ZERO_PADDED / DGA_ROWS_B_ZERO_PADDED (dga_cast_to_fp8_*_ld writes
 * such rows; make the stride a multiple of 128 where possible -- rows that start inside a cache line cost two line requests per
 * 128-byte piece and give the gain back, profiles/r04_odd_k_rows.txt); an operand without the promise goes through the padding
 * pass ALONE -- weights padded once at load time leave only
 * the activations to re-lay out, and operands that both come from the _ld quantisers need no pass at all (the reference fuses
 * its re-layout into the matmul launch instead: op_kernel/kernel/padding_common_matmul_kernel.h:33-107).
 * DGA_E_ALIGN: a stride that is neither K nor a multiple of 16; DGA_E_SHAPE: a stride below K. */
#define DGA_ROWS_A_ZERO_PADDED 1
#define DGA_ROWS_B_ZERO_PADDED 2
int dga_gemm_fp8_fp8_bf16_nt_strided(const void *a, int64_t lda, const float *sfa, const void *b, int64_t ldb, const float *sfb,
                                     void *out, int m, int n, int k, int flags, const dga_tiling_t *tiling, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* m_grouped_gemm_fp8_fp8_bf16_nt_masked: G independent problems
 *   a [G,m_max,K], sfa [G,m_max,KB], b [G,N,K], sfb [G,NB,KB], out [G,m_max,N];
 *   only rows < masked_m[g] (device int32[G]) of out[g] are written.  masked_m is read on the device while the call
 *   executes (like every operand it must not be written concurrently).
 * No reference counterpart beyond the uniform batch loop (generate_code.hpp:149-153); SURVEY.md 8(b). */
int dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked(const void *a, const float *sfa, const void *b, const float *sfb,
                                              void *out, const int32_t *masked_m, int groups, int m_max, int n,
                                              int k, int expected_m, const dga_tiling_t *tiling, void *workspace,
                                              size_t workspace_bytes, void *stream);

/* The masked grouped GEMM on rows that stay where they are: instead of a packed [G, m_max, K] activation tensor, row r of
 * group g is row row_index[g * m_max + r] of ONE flat source -- a (rows x lda bytes, lda % 16 == 0, rows * lda < 2 GiB),
 * its scales at sfa + row * sfa_ld floats (they may ride inside the same payload rows) -- and its result goes to row
 * row_index[g * m_max + r] of the flat destination out (ldc bf16 elements per row).  Only entries r < masked_m[g] of the
 * table are read.  This removes the pack / unpack copies either side of the GEMM in the expert-sharded forward (the
 * tokens are gathered by the kernel's own tile loads and scattered by its stores); dga_route_slots builds the table.
 * K % 16 != 0 runs the element-wise kernel.  No reference counterpart (row 8(e)). */
int dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed(const void *a, int64_t lda, const float *sfa, int64_t sfa_ld,
                                                      const void *b, const float *sfb, void *out, int64_t ldc,
                                                      const int64_t *row_index, int64_t rows, const int32_t *masked_m,
                                                      int groups, int m_max, int n, int k, int expected_m,
                                                      const dga_tiling_t *tiling, void *workspace, size_t workspace_bytes,
                                                      void *stream);

/* The aclnn operator in its own dtypes (CatlassDynamicMatmul, op_host/catlass_dynamic_matmul.cpp:50-80; device entry
 * op_kernel/catlass_dynamic_matmul.cpp:16-45): out[M,N] = self[M,K] . mat2, self row-major, mat2 logical [K,N] stored
 * column-major = physical [N,K] (NT, catlass_dynamic_matmul_tiling.cpp:83-84), all three of `dtype` (DGA_DT_FP16 or
 * DGA_DT_BF16; fp32 accumulate, one RNE rounding at the end).  workspace: dga_catlass_dynamic_matmul_workspace_bytes()
 * (padded operand copies when K % 64 != 0 or a base is not 16-byte aligned; split-K slabs for small M); NULL is legal
 * (single pass, element-wise kernel for unaligned shapes). */
size_t dga_catlass_dynamic_matmul_workspace_bytes(int m, int n, int k, const void *self, const void *mat2);
int dga_catlass_dynamic_matmul(const void *self, const void *mat2, void *out, int m, int n, int k, int dtype,
                               void *workspace, size_t workspace_bytes, void *stream);

/* m_grouped_gemm_fp8_fp8_bf16_nt_contiguous: the prefill-side MoE layout.
 *   a [m_sum,K], sfa [m_sum,KB], b [G,N,K], sfb [G,NB,KB], out [m_sum,N], m_indices device int32[m_sum].
 * Row r is multiplied with b[m_indices[r]]; rows with m_indices[r] < 0 are padding and are not written.
 * Layout contract: the rows of one group are consecutive, every group segment starts at a multiple of
 * DGA_CONTIGUOUS_M_ALIGNMENT rows, and a segment's padding rows (-1) follow its valid rows.
 * No reference counterpart (the reference has only the uniform batch loop, generate_code.hpp:149-153). */
int dga_m_grouped_gemm_fp8_fp8_bf16_nt_contiguous(const void *a, const float *sfa, const void *b, const float *sfb,
                                                  void *out, const int32_t *m_indices, int m_sum, int groups, int n,
                                                  int k, const dga_tiling_t *tiling, void *workspace,
                                                  size_t workspace_bytes, void *stream);

/* Quantisers upstream of the GEMM: x [rows,k] contiguous, x_dtype in
 * {DGA_DT_FP32, DGA_DT_BF16, DGA_DT_FP16} -> q [rows,k] e4m3fn bytes and fp32 scales
 *   1x128:    sf [rows, ceil(k/128)]              (the A-operand / activation format)
 *   128x128:  sf [ceil(rows/128), ceil(k/128)]    (the B-operand / weight format)
 * scale = amax/448 (1 when the block is all zero), q = RNE-satfinite(x / scale).  The reference's inputs are fp16
 * files from numpy (scripts/gen_data.py:10-30); it has no quantiser. */
int dga_cast_to_fp8_1x128(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, float *sf, void *stream);
int dga_cast_to_fp8_128x128(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, float *sf, void *stream);
/* The same writing rows ldq bytes apart (k <= ldq <= 128 * ceil(k/128)) with ZEROS from byte k to the end of each row: with
 * ldq = round_up(k, 128) (whole cache lines; round_up(k, 16) is the minimum) the result is an operand
 * dga_gemm_fp8_fp8_bf16_nt_strided reads in place (DGA_ROWS_*_ZERO_PADDED). */
int dga_cast_to_fp8_1x128_ld(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, void *stream);
int dga_cast_to_fp8_128x128_ld(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, void *stream);
/* The same with flags (ABI 6).  DGA_CAST_UE8M0: every block scale is rounded UP to a power of two, 2^ceil(log2(amax / 448)) -- the
 * "UE8M0" scales of upstream DeepGEMM's per_token_cast_to_fp8(..., use_ue8m0=True); still written as fp32.  GEMM calls on such
 * operands may set DGA_POLICY_UE8M0_SCALES (the scales then ride in the matrix instruction's E8M0 operands).  ldq as in the _ld forms
 * (ldq = k: contiguous rows).  DGA_E_RANGE: an unknown flag. */
#define DGA_CAST_UE8M0 1
int dga_cast_to_fp8_1x128_ex(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, int flags, void *stream);
int dga_cast_to_fp8_128x128_ex(const void *x, int x_dtype, int64_t rows, int64_t k, void *q, int64_t ldq, float *sf, int flags, void *stream);

/* ---- the framework's 28-int Config (deep_gemm_ascend/framework/csrc/jit/get_best_config.hpp) ---- */

/* struct Config in declaration order (get_best_config.hpp:12-31), 28 uint32. */
int dga_get_best_config(uint32_t batch, uint32_t m, uint32_t n, uint32_t k, uint32_t out[28]);
int dga_get_bench_config(uint32_t m, uint32_t n, uint32_t k, uint32_t m_sections, uint32_t n_sections,
                         uint32_t m_sec_o_blocks, uint32_t n_sec_o_blocks, uint32_t k_o_iter_blocks,
                         uint32_t db_o_blocks, uint32_t out[28]);
/* run_mmad_bench's host write-back: slots 0..5 are the knobs in, slots 6..27 are filled in the
 * Python-binding order of gemm_bench.hpp:68-81 (m,n,k,batch,k_iters,...,r_db_num). */
int dga_bench_params_fill(uint32_t m, uint32_t n, uint32_t k, int32_t params[28]);
/* benchmark_msprof's order (benchmark_util.h:78-85): m,n,k,6 knobs,batch,k_iters,... */
int dga_bbit_params(uint32_t m, uint32_t n, uint32_t k, uint32_t m_sections, uint32_t n_sections,
                    uint32_t m_sec_o_blocks, uint32_t n_sec_o_blocks, uint32_t k_o_iter_blocks,
                    uint32_t db_o_blocks, uint32_t out[28]);

/* ---- the framework's 16-bit path (what the reference ships today) ----------------------- */

/* run_mmad_rtc (python_api.cpp:18, gemm.hpp:68-111): z[B,M,N] f32 = x[B,M,K] . y[B,K,N], x/y bf16 or fp16. */
int dga_run_mmad_rtc(const void *x, const void *y, float *z, int batch, int m, int n, int k, int dtype,
                     void *stream);
/* run_mmad_bench (python_api.cpp:23, gemm_bench.hpp:49-113): z[M,N] f32 = x[M,K] . y[K,N];
 * params_host = the 28 ints after dga_bench_params_fill (the knobs only steer the Ascend kernel). */
int dga_run_mmad_bench(const void *x, const void *y, float *z, int m, int n, int k, int dtype,
                       const int32_t *params_host, void *stream);

/* The same with a caller-provided workspace of dga_mmad_workspace_bytes(): y is transposed (and x padded when K is not
 * a multiple of 64) into it and the LDS-DMA / MFMA tile kernel runs; without a workspace the entry points above gather
 * fragments straight from global memory (same results, much slower). */
size_t dga_mmad_workspace_bytes(int batch, int m, int n, int k, const void *x);
int dga_run_mmad_rtc_ws(const void *x, const void *y, float *z, int batch, int m, int n, int k, int dtype,
                        void *workspace, size_t workspace_bytes, void *stream);
int dga_run_mmad_bench_ws(const void *x, const void *y, float *z, int m, int n, int k, int dtype,
                          const int32_t *params_host, void *workspace, size_t workspace_bytes, void *stream);

/* ---- expert sharding helper (the reference has no routing or collective of any kind) ------- */

/* Token routing for the dispatch: counts[g] = number of t with expert_ids[t] == g (int64, zeroed here) and pos[t] =
 * position of token t in the expert-sorted order (an expert's tokens are contiguous, their relative order is
 * unspecified; ids outside [0, groups) get pos -1 and are not counted).  One pass of atomics in place of a device sort
 * and a histogram.  No reference counterpart (row 8(e)). */
int dga_route_tokens(const int64_t *expert_ids, int64_t tokens, int groups, int64_t *counts, int64_t *pos, void *stream);

/* Capacity-bounded slot assignment, the routing step of the sharded forward that needs no host round trip (static
 * shapes: the whole forward is capturable in a HIP graph).  Row r carries an int32 key at keys + r * key_stride_bytes
 * (an expert id; negative = unused row).  hi = key / key_div, lo = key % key_div;
 *   bucket = hi                                        (key_sub == 0)
 *   bucket = (lo / key_sub) * key_mul + hi             (key_sub > 0: expert chunk, then destination rank)
 * dest[r] = bucket * cap + (next free slot of the bucket, one atomic), or -1 for an unused row / an unknown bucket; a
 * full bucket gives -1 and adds 1 to *overflow per dropped row (device int32 counter, sticky -- the caller reads and clears it).
 * counts (device int32[buckets]) end as the rows placed per bucket (zeroed first when zero_counts != 0); tags != NULL:
 * the int32 at tags + dest[r] * tag_stride_bytes receives lo (the payload header the receiving rank routes by).
 * Slot order inside a bucket is the atomics' arrival order.  No reference counterpart (row 8(e)).
 * inverse != NULL (device int64[buckets * cap]): inverse[dest[r]] = r + inverse_base, the slot -> row table of
 * dga_m_grouped_gemm_fp8_fp8_bf16_nt_masked_indexed. */
int dga_route_slots(const void *keys, int64_t key_stride_bytes, int64_t rows, int key_div, int key_sub, int key_mul,
                    int buckets, int cap, int32_t *counts, int zero_counts, int64_t *dest, void *tags,
                    int64_t tag_stride_bytes, int32_t *overflow, int64_t *inverse, int64_t inverse_base, void *stream);

/* Indexed row copy on the device: for r in [0, rows):
 *   dst[(dst_index ? dst_index[r] : r) * dst_row_stride .. +row_bytes) = src[(src_index ? src_index[r] : r) * src_row_stride ..)
 * (strides in bytes; index arrays are device int64).  Packs token rows for the dispatch all-to-all, scatters the
 * received rows into the masked [G, m_max, K] layout, and the reverse for the combine. */
int dga_copy_rows(void *dst, int64_t dst_row_stride, const int64_t *dst_index, const void *src, int64_t src_row_stride,
                  const int64_t *src_index, int64_t row_bytes, int64_t rows, void *stream);
/* Two row streams with shared indices in one launch: dst0[di] = src0[si] (row_bytes0) and dst1[di] = src1[si]
 * (row_bytes1) -- the dispatch's pack (fp8 bytes + scales -> one payload row) and unpack. */
int dga_copy_rows2(void *dst0, int64_t dst0_row_stride, const void *src0, int64_t src0_row_stride, int64_t row_bytes0,
                   void *dst1, int64_t dst1_row_stride, const void *src1, int64_t src1_row_stride, int64_t row_bytes1,
                   const int64_t *dst_index, const int64_t *src_index, int64_t rows, void *stream);

/* ---- the expert-sharded forward behind the C ABI (csrc/dga_sharded.cpp) ----------------------------------
 * What a C++ host (the reference's host language: framework/csrc/python_api.cpp) calls to run BASELINE configs[4]: expert g
 * lives on rank g / (groups_total / world), one all-to-all of payload rows each way, every shape static, nothing read back.
 * The library computes the buffer layout, the fixed step sequence ("plan") and executes it on the caller's streams; the two
 * collectives are the caller's (a callback), so the library does not link RCCL.  The reference has no counterpart ("multi-card"
 * = independent processes: benchmark_msprof/main.cpp:24-26, framework/benchmark/benchmark.py:249-253). */
typedef struct dga_sharded_shape_t {
    int32_t world, rank;
    int32_t groups_total;     /* experts over all ranks (a multiple of world) */
    int32_t m_max, n, k;      /* rows per expert (capacity), output columns, reduction length */
    int32_t chunks;           /* a rank's experts are processed in this many chunks, whose dispatch / GEMM / combine overlap on
                                 three streams; <= 0: 2 when world > 1 and the experts per rank allow it, else 1 */
    int32_t max_tokens;       /* the largest number of tokens one rank brings to a forward (sizes the exchange slices: the same
                                 on every rank); <= 0: groups_local * m_max */
    float capacity_factor;    /* rows reserved per (chunk, destination rank) = this x the even share, rounded up to 16;
                                 <= 0: the provable bound min(max_tokens, experts per chunk x m_max) */
    int32_t indexed;          /* 1: the GEMM gathers its rows from the receive buffer and scatters results into the buffer that
                                 travels back (no unpack / gather copy); 0: packed masked layout */
    int32_t policy;           /* dispatchPolicyTag of the GEMM (DGA_POLICY_STRICT, DGA_POLICY_BF16_EXACT, ...); -1 = the library's default
                                 arithmetic (dga_default_policy); -3 = the fast policy's tiling | DGA_POLICY_UE8M0_SCALES; -2 = the fast policy's
                                 own tiling, schedule included */
} dga_sharded_shape_t;

/* Payload row of the dispatch exchange: [K fp8 bytes][ceil(K/128) fp32 scales][int32 header = expert index on its owner, -1 =
 * unused row], padded to a multiple of 128 bytes.  Exchange buffers hold chunks x world x pair_capacity rows; a rank sends
 * peer p the rows [c * rows_per_chunk + p * pair_capacity, + pair_capacity) of chunk c (EQUAL splits: a static shape). */
typedef struct dga_sharded_layout_t {
    int32_t groups_local, groups_per_chunk, chunks, kb, nb;
    int32_t indexed;          /* the shape's flag after the limits are applied (32-bit tile offsets, K % 4) */
    int64_t hdr_offset, row_bytes, pair_capacity, rows_per_chunk, rows_total, max_tokens;
    uint64_t send_bytes, recv_bytes;       /* payload rows out / in */
    uint64_t osend_bytes, oback_bytes;     /* bf16 [rows_total, n] result rows out / back */
    uint64_t slot_bytes;                   /* int64 [max_tokens]: slot of every token */
    uint64_t rdest_bytes;                  /* int64 [rows_total]: row of every received row in the masked layout */
    uint64_t row_of_slot_bytes;            /* int64 [groups_local * m_max]: slot -> source row (indexed only) */
    uint64_t pair_cnt_bytes;               /* int32 [chunks * world] */
    uint64_t masked_m_bytes;               /* int32 [groups_local]: rows every local expert received (the GEMM's masked_m) */
    uint64_t packed_a_bytes, packed_sfa_bytes, packed_out_bytes;   /* the masked layout [Gl, m_max, .] (packed path only) */
    int32_t events;           /* events the executor needs (dga_sharded_events_create) */
    int32_t steps;            /* length of the plan */
} dga_sharded_layout_t;

typedef struct dga_sharded_buffers_t {
    void *send, *recv, *osend, *oback;
    int64_t *slot, *rdest, *row_of_slot;
    int32_t *pair_cnt, *masked_m;
    int32_t *overflow;        /* sticky device counter of rows dropped by a full expert / pair slice (the caller reads and clears it) */
    void *packed_a; float *packed_sfa; void *packed_out;
    const void *b;            /* resident weights of this rank: [groups_local, n, k] e4m3fn */
    const float *sfb;         /* [groups_local, ceil(n/128), ceil(k/128)] */
    void *workspace; size_t workspace_bytes;   /* dga_workspace_bytes of the grouped tiling (may be NULL) */
} dga_sharded_buffers_t;

/* Steps of the plan.  stream: 0 = the caller's, 1 = dispatch side stream, 2 = combine side stream. */
enum { DGA_STEP_WAIT_EVENT = 0,          /* stream waits for event */
       DGA_STEP_RECORD_EVENT = 1,        /* event recorded on stream */
       DGA_STEP_CLEAR_HEADERS = 2,       /* header of every send row = -1 */
       DGA_STEP_ROUTE_SOURCE = 3,        /* dga_route_slots on the tokens' expert ids (+ header tags) */
       DGA_STEP_PACK = 4,                /* dga_copy_rows2: token bytes + scales into their slots */
       DGA_STEP_ZERO_COUNTS = 5,         /* masked_m = 0 */
       DGA_STEP_ZERO_DROPPED = 6,        /* result rows of the tokens that found no slot = 0 (nothing else ever writes them) */
       DGA_STEP_ALL_TO_ALL_DISPATCH = 7, /* the callback, direction 0: payload rows of one chunk */
       DGA_STEP_ROUTE_RECEIVED = 8,      /* dga_route_slots keyed by the received headers: masked_m and the row tables */
       DGA_STEP_UNPACK = 9,              /* packed path: received rows into the masked layout */
       DGA_STEP_GEMM = 10,               /* the grouped masked-M GEMM of a chunk's experts */
       DGA_STEP_GATHER_OUT = 11,         /* packed path: result rows into the buffer that travels back */
       DGA_STEP_ALL_TO_ALL_COMBINE = 12, /* the callback, direction 1: bf16 result rows of one chunk */
       DGA_STEP_RESTORE_ORDER = 13,      /* result[t] = returned row slot[t] */
       DGA_STEP_ZERO_UNROUTED = 14 };    /* returning rows of the received rows that found no place in their expert = 0 */
typedef struct dga_sharded_step_t {
    int32_t op, stream, chunk, event;
    int64_t row_begin, rows;             /* slice of the exchange buffers */
    int32_t group_begin, groups;         /* local experts of the step */
} dga_sharded_step_t;

/* The collective: send peer p the `bytes_per_peer` bytes at send + p * bytes_per_peer, receive its slice at recv + p *
 * bytes_per_peer, asynchronously on `stream` (ncclGroupStart; ncclSend/ncclRecv x world; ncclGroupEnd -- or ncclAllToAll).
 * direction 0 = dispatch, 1 = combine.  Returns 0 on success. */
typedef int (*dga_all_to_all_fn)(void *user, int direction, int chunk, const void *send, void *recv, size_t bytes_per_peer,
                                 void *stream);

int dga_sharded_layout(const dga_sharded_shape_t *shape, dga_sharded_layout_t *out);
/* steps == NULL: *count = the plan's length.  Pure host arithmetic (no device is touched). */
int dga_sharded_plan(const dga_sharded_shape_t *shape, dga_sharded_step_t *steps, int capacity, int *count);
/* One forward: tok_q [tokens, k] e4m3fn, tok_sf [tokens, ceil(k/128)] f32, expert_ids int64 [tokens] (global expert of every
 * token), all device memory -> result bf16 [tokens, n] in token order.  streams[3]: hipStream_t (they may all be the same one);
 * events: layout.events handles (dga_sharded_events_create).  Asynchronous; nothing is read back; capturable in a HIP graph.
 * world == 1: no exchange, all_to_all and events may be NULL. */
int dga_sharded_forward(const dga_sharded_shape_t *shape, const dga_sharded_buffers_t *buffers, const void *tok_q,
                        const float *tok_sf, const int64_t *expert_ids, int tokens, void *result, int expected_m,
                        void *const *streams, void *const *events, dga_all_to_all_fn all_to_all, void *user);
/* hipEvent_t handles (timing disabled) for the executor: host objects, no device memory. */
int dga_sharded_events_create(int count, void **events);
int dga_sharded_events_destroy(int count, void **events);

/* ---- diagnostics ------------------------------------------------------------------------- */

/* The shader clock held inside the dense kernel's main loop ("record the measured clock" beside the
 * vendor peak).  Runs `launches` back-to-back launches of the loop-clock build of the kernel `tiling` selects -- the
 * product kernel plus one s_memtime / s_memrealtime pair either side of the k loop -- synchronises `stream`, and returns
 * the median over waves of shader ticks / 100 MHz ticks (clock_mhz) and of the loop's duration (loop_us, nullable).
 * Compiled for the kernels of BASELINE configs[1] / [2] (256x256 continuous, 128x256 3-stage with loader waves); other tilings,
 * split-K, K % 128 != 0: DGA_E_TILING.  scratch: device memory, 16 bytes per wave (128 per tile).  The reference times
 * with msprof from outside (framework/benchmark/benchmark.py:400-418) and has no counterpart. */
int dga_gemm_fp8_loop_clock(const void *a, const float *sfa, const void *b, const float *sfb, void *out, int m, int n,
                            int k, const dga_tiling_t *tiling, void *scratch, size_t scratch_bytes, int launches,
                            void *stream, float *clock_mhz, float *loop_us);

/* What the matrix pipe of this device sustains with the operands already in registers: `launches` back-to-back launches of
 * a loop of the policy's matrix instruction with the fp32 promotion beside it (mode 0: v_mfma_scale_f32_16x16x128_f8f6f4 + 4
 * FMAs -- the fast path; mode 1: 4 chained v_mfma_f32_16x16x32_bf16 + 4 FMAs + 8 e4m3 -> bf16 conversions -- the bf16-exact
 * policy's 64 x 64 wave tile), two waves per SIMD on every CU, random e4m3 bytes; *tflops = the last launch's rate.  The
 * ceiling bench.py prices the product kernels against beside the vendor peak (`roofline.ceiling_tflops`).
 * scratch: device memory, >= 16 KiB + 2 KiB per CU.  A diagnostic (it synchronises); no reference counterpart. */
int dga_mfma_ceiling(int mode, int launches, void *scratch, size_t scratch_bytes, void *stream, float *tflops);

/* ---- misc -------------------------------------------------------------------------------- */
const char *dga_status_string(int status);
int dga_last_hip_error(void);
int dga_abi_version(void);
/* fills coreNum etc. from hipDeviceProp of the current device */
int dga_device_platform(dga_platform_t *out);

#ifdef __cplusplus
}
#endif
#endif /* DGA_HIP_H */
