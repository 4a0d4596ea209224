// The compiled menu of fp8 tile-kernel builds, shared between the launcher (dga_launch.hip, which only takes the
// addresses of the launch functions) and the translation units that instantiate the kernels
// (dga_launch_menu_{a,b,c}.hip -- split so that `make -j` compiles the menu in parallel).  This is the AOT replacement
// of the reference's per-shape cmake-subprocess JIT (/root/reference/deep_gemm_ascend/framework/csrc/jit/compiler.hpp:26-93).
#pragma once
#include "dga_internal.hpp"
#include "gemm_fp8_kernel.hpp"

namespace dga {

// Launch one build on `stream`: grid from p (launch_tiles / groups x tiles), LDS attribute set once per device.
// K % 128 != 0 takes the instantiation with the per-lane beyond-K test in its DMA slots.  CLK: the loop-clock
// diagnostic build (K % 128 == 0 only).
template <class Cfg, int PP, bool CLK = false>
int launch_cfg(const GemmParams &p, hipStream_t stream);

// X(BM, BN, WM, WN, STAGES, PP)
#define DGA_MENU_A(X) X(256, 256, 4, 2, 2, 0) X(256, 256, 4, 2, 2, 1) X(256, 256, 4, 2, 2, 2)
#define DGA_MENU_B(X)                                                                                              \
    X(128, 256, 2, 2, 2, 2) X(256, 128, 4, 1, 2, 2) X(128, 128, 2, 2, 2, 2) X(64, 256, 1, 4, 2, 2) X(128, 256, 2, 4, 2, 2)
#define DGA_MENU_C(X)                                                                                              \
    X(128, 256, 2, 2, 2, 0) X(256, 128, 4, 1, 2, 0) X(128, 128, 2, 2, 2, 0) X(64, 256, 1, 4, 2, 0) X(64, 128, 1, 4, 2, 0) \
    X(128, 256, 2, 4, 2, 0) X(128, 256, 2, 4, 3, 0) X(128, 256, 2, 2, 3, 0) X(128, 128, 2, 2, 3, 0) X(64, 256, 1, 4, 3, 0) \
    X(32, 256, 1, 4, 2, 0) X(32, 128, 1, 4, 2, 0) X(16, 256, 1, 4, 2, 0) X(16, 128, 1, 4, 2, 0)                        \
    X(64, 128, 1, 4, 3, 0) X(32, 256, 1, 4, 3, 0) X(32, 128, 1, 4, 3, 0) X(16, 256, 1, 4, 3, 0) X(16, 128, 1, 4, 3, 0)
// loader-wave build (GemmCfg<..., LCW = 4>, dispatchPolicyTag 4): the masked grouped weight stream and dense problems
// that give every CU one 128x256 tile
#define DGA_MENU_LC(X) X(128, 256, 2, 2, 3, 0) X(128, 128, 2, 2, 3, 0) X(64, 256, 1, 4, 3, 0) X(64, 128, 1, 4, 3, 0) X(16, 128, 1, 4, 3, 0)
// loop-clock builds: the kernels of BASELINE configs[1] and configs[2]
#define DGA_MENU_CLK(X) X(256, 256, 4, 2, 2, 2)
#define DGA_MENU_CLK_LC(X) X(128, 256, 2, 2, 3, 0)   // configs[2] runs the loader-wave build

#define DGA_MENU_EXTERN(BM, BN, WM, WN, ST, PP) \
    extern template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST>, PP, false>(const GemmParams &, hipStream_t);
#define DGA_MENU_EXTERN_CLK(BM, BN, WM, WN, ST, PP) \
    extern template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST>, PP, true>(const GemmParams &, hipStream_t);
DGA_MENU_A(DGA_MENU_EXTERN)
DGA_MENU_B(DGA_MENU_EXTERN)
DGA_MENU_C(DGA_MENU_EXTERN)
DGA_MENU_CLK(DGA_MENU_EXTERN_CLK)
#define DGA_MENU_EXTERN_CLK_LC(BM, BN, WM, WN, ST, PP) \
    extern template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST, 4>, PP, true>(const GemmParams &, hipStream_t);
DGA_MENU_CLK_LC(DGA_MENU_EXTERN_CLK_LC)
#define DGA_MENU_EXTERN_LC(BM, BN, WM, WN, ST, PP) \
    extern template int launch_cfg<GemmCfg<BM, BN, WM, WN, ST, 4>, PP, false>(const GemmParams &, hipStream_t);
DGA_MENU_LC(DGA_MENU_EXTERN_LC)

// persistent loader-wave builds (gemm_fp8_persistent_kernel.hpp, dispatchPolicyTag 5; dga_launch_menu_d.hip): one workgroup
// per CU walks its share of the tiles, the LDS ring runs across tile boundaries.  Every loader-wave tile has one.
template <class Cfg>
int launch_persistent(const GemmParams &p, hipStream_t stream);
#define DGA_MENU_EXTERN_PS(BM, BN, WM, WN, ST, PP) \
    extern template int launch_persistent<GemmCfg<BM, BN, WM, WN, ST, 4>>(const GemmParams &, hipStream_t);
DGA_MENU_LC(DGA_MENU_EXTERN_PS)

// bf16-exact builds (gemm_fp8_kernel.hpp MATH = 1, dispatchPolicyTag 7; dga_launch_menu_e.hip): the e4m3 bytes up-converted to
// bf16 in registers, a scale block = four chained v_mfma_f32_16x16x32_bf16.  Wave tiles of at most 64 x 64 (the bf16 A
// fragments and a double-buffered bf16 B fragment live beside the accumulators), three LDS stages.
template <class Cfg>
int launch_bf16x(const GemmParams &p, hipStream_t stream);
#define DGA_MENU_BX(X) X(128, 256, 2, 4, 3, 0) X(128, 128, 2, 2, 3, 0) X(64, 256, 1, 4, 3, 0) X(64, 128, 1, 4, 3, 0) X(32, 128, 1, 4, 3, 0)
#define DGA_MENU_EXTERN_BX(BM, BN, WM, WN, ST, PP) \
    extern template int launch_bf16x<GemmCfg<BM, BN, WM, WN, ST>>(const GemmParams &, hipStream_t);
DGA_MENU_BX(DGA_MENU_EXTERN_BX)

// image builds of the bf16-exact policy (gemm_fp8_bf16x_image_kernel.hpp; dga_launch_menu_f.hip): 128 x 256 tile, both operands
// converted once per workgroup into a bf16 LDS image; waves = 8 (two per SIMD, 64 x 64 wave tiles) or 4 (one per SIMD, 64 x 128).  Dense and masked-grouped rasters (split-K too);
// DGA_E_TILING for the contiguous / indexed layouts
int launch_bf16x_image(const GemmParams &p, int waves, hipStream_t stream);
// the persistent form of the 128 x 256 in-register build (dense / masked grouped rasters); DGA_E_TILING: not a launch it takes
int launch_bf16x_persistent(const GemmParams &p, hipStream_t stream);
// the masked grouped layout's kernel (gemm_fp8_bf16x_grouped_kernel.hpp; dga_launch_menu_l.hip): two k blocks in flight, per-m-tile row skipping
int launch_bf16x_grouped(const GemmParams &p, hipStream_t stream);

// loader-wave build of the 128 x 256 tile for rows that start at any byte (K % 16 != 0, no padded copy; dga_launch_menu_d.hip)
int launch_unaligned(const GemmParams &p, hipStream_t stream);

// one-launch workgroup split-K for dense problems of at most 64 rows (gemm_fp8_wsk_kernel.hpp; dga_launch_menu_g.hip; kernelSerial
// DGA_KERNEL_SPLITK_WORKGROUP): bit-identical to the two-launch split-K with splitkFactor 8.  DGA_E_TILING for anything else
int launch_wsk(const GemmParams &p, hipStream_t stream);
// the LDS-DMA staged builds (M <= 32); math 1 = the bf16-exact policy's arithmetic; DGA_E_TILING: not a problem they take
int launch_wsk_dma(const GemmParams &p, hipStream_t stream, int math = 0);
int wsk_rows(int m);
int wsk_max_ntiles(int m);

// hardware-scale builds (gemm_fp8_kernel.hpp MATH = 2; dga_launch_menu_i.hip): block scales that are exact powers of two ride in
// the matrix instruction's E8M0 operands, the MFMA accumulates in place.  DGA_E_TILING: no such build of that tile
int launch_ue8m0(int bm, int bn, bool loaders, bool cont, const GemmParams &p, hipStream_t stream);
// ... and the 256 x 256 tile on FOUR waves (wave tile 128 x 128, accumulators in AGPRs; dga_launch_menu_j.hip), continuous loop
int launch_ue8m0_w4(const GemmParams &p, hipStream_t stream);
// bf16-exact arithmetic for power-of-two scales (MATH = 3: scales folded into the A conversions, the bf16 MFMA accumulates in place, AGPR
// accumulators; dga_launch_menu_j.hip).  DGA_E_TILING: no such build of that tile
int launch_bf16u(int bm, int bn, const GemmParams &p, hipStream_t stream);

// one-launch Stream-K build of the 256 x 256 continuous kernel (gemm_fp8_streamk_kernel.hpp; dga_launch_menu_k.hip): dense rasters of
// full tiles, fp32 partial tiles through the caller's workspace.  DGA_E_TILING: not a problem it takes
int launch_streamk(const GemmParams &p, void *ws, size_t ws_bytes, bool ue8m0, hipStream_t stream);
size_t streamk_workspace_bytes();
// ... and of the bf16-exact policy's persistent 128 x 256 kernel (gemm_fp8_bf16x_streamk_kernel.hpp; dga_launch_menu_m.hip): any dense
// raster with a partial last round.  DGA_E_TILING: not a launch it takes (nothing to cut, no workspace, co-residency not guaranteed)
int launch_bf16x_streamk(const GemmParams &p, void *ws, size_t ws_bytes, hipStream_t stream);
size_t bx_streamk_workspace_bytes();
// workgroups a launch on this stream can count on being resident together, one per CU; 0 when a CU mask narrows the queue
int coresident_workgroups(hipStream_t stream);
// one-launch split-K for decode rows under the bf16-exact policy (gemm_fp8_bf16x_dsk_kernel.hpp; dga_launch_menu_n.hip): dense problems of
// at most as many 64 x 128 tiles as CUs; `splits` = the tiling's splitkFactor.  DGA_E_TILING: not a launch it takes
int launch_bf16x_dsk(const GemmParams &p, int splits, void *ws, size_t ws_bytes, hipStream_t stream);
int bx_dsk_splits(int64_t tiles, int kb, int want, int cus);
size_t bx_dsk_workspace_bytes(int64_t tiles, int splits);

// persistent continuous-pipeline build of the 256x256 tile (gemm_fp8_cont_persistent_kernel.hpp, dispatchPolicyTag 6): dense
// rasters of full tiles only -- launch_cont_persistent returns DGA_E_TILING for anything else
int launch_cont_persistent(const GemmParams &p, hipStream_t stream);

}  // namespace dga
