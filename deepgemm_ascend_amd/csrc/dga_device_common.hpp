// Device-side building blocks shared by the fp8 and the 16-bit tile kernels: vector types, the tile configuration,
// the LDS image swizzle, buffer descriptors and the LDS-DMA issue helpers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace dga {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef __bf16 v2bf __attribute__((ext_vector_type(2)));

// LCW > 0 ("loader / consumer"): the workgroup carries LCW EXTRA waves that do nothing but issue the LDS-DMA, so the
// computing waves' instruction streams hold no refill at all (plain loop only); the refill of a stage goes out in one burst
// the moment the barrier releases it.
template <int BM, int BN, int WM, int WN, int ST = 2, int LCW = 0>
struct GemmCfg {
    static constexpr int kBM = BM, kBN = BN, kWM = WM, kWN = WN;
    static constexpr bool kLC = LCW > 0;
    static constexpr int NT = (WM * WN + LCW) * 64;
    static constexpr int TM = BM / WM / 16;  // m-tiles (16 rows) per wave
    static constexpr int TN = BN / WN / 16;  // n-tiles per wave (even)
    // Waves that issue the LDS-DMA.  (Giving all of it to the first-dispatched half of an 8-wave workgroup -- the
    // half that wins every MFMA arbitration -- evens the two halves out but measured 2-4 % slower overall, r01.)
    static constexpr int DMA_WAVES = LCW > 0 ? LCW : WM * WN;
    static constexpr int DNT = DMA_WAVES * 64;  // threads that issue DMA
    static constexpr int A_ROWS = BM * 8 >= DNT ? BM : DNT / 8;  // tiny BM: pad the image to whole wave-instructions
    static constexpr int A_BYTES = A_ROWS * 128;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int SC_SLOTS = ((BM + 8 + DNT - 1) / DNT) * DNT;  // sfa rows, then sfb entries, padded
    static constexpr int SC_BYTES = SC_SLOTS * 4;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES + SC_BYTES;
    static constexpr int STAGES = ST;  // LDS stages (3 only with the PP = 0 loop)
    static constexpr int LDS_BYTES = STAGES * STAGE_BYTES;
    static constexpr int A_ITERS = A_ROWS * 8 / DNT;
    static constexpr int B_ITERS = BN * 8 / DNT;
    static constexpr int SC_ITERS = SC_SLOTS / DNT;
    static constexpr int LOADS_PER_STAGE = A_ITERS + B_ITERS + SC_ITERS;
    static_assert(BM % (WM * 16) == 0 && BN % (WN * 32) == 0, "wave tile");
    static_assert(BN % 128 == 0 && BN / WN <= 128 && 128 % (BN / WN) == 0, "a wave's n range lies in one 128-wide scale block");
    static_assert((A_ROWS * 8) % DNT == 0 && (BN * 8) % DNT == 0, "whole wave-instructions per tile");
    static_assert(BN / 128 + (BN % 128 != 0) <= 8, "sfb slots");
};

// LDS image: row r of a tile is 128 bytes = 8 chunks of 16 B; chunk c is stored at
// chunk position c ^ x(r).  x is chosen per operand so that the 16-lane groups of
// ds_read_b128 touch 16 distinct 16-byte slots of the 256-byte bank row.
__device__ __forceinline__ int swz_a(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ int swz_b(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;

// Buffer descriptor (raw, stride 0): base address, byte extent, DATA_FORMAT=32 flags word as in the guide's T8.
__device__ __forceinline__ v4i make_rsrc(const void *base, int64_t bytes)
{
    const uint64_t b = (uint64_t)(uintptr_t)base;
    const uint32_t n = bytes > 0x7FFFFFFFll ? 0x7FFFFFFFu : (uint32_t)bytes;
    v4i r;
    r.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    r.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)((b >> 32) & 0xFFFFu));
    r.z = __builtin_amdgcn_readfirstlane((int)n);
    r.w = 0x00020000;
    return r;
}

// LDS-DMA, 16 B per lane: LDS[m0 + 16*lane] = buffer[voff + soff .. +16).  M0 is written in the same statement
// that uses it (the descriptor is built once, long before the first use: no readfirstlane hazard to cover).
__device__ __forceinline__ void dma16(uint32_t voff, v4i rsrc, uint32_t soff, uint32_t lds_addr)
{
#if defined(DGA_ABL_DMA_EXEC0)   // diagnostic: the instruction issues with every lane off (no TA / LDS / memory work)
    unsigned long long keep;
    asm volatile("s_mov_b64 %4, exec\n\ts_mov_b64 exec, 0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "buffer_load_dwordx4 %0, %1, %2 offen lds\n\ts_mov_b64 exec, %4"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr), "s"(keep) : "memory");
#elif defined(DGA_ABL_DMA_OOB)   // diagnostic: every lane out of range (TA + LDS zero-fill, no memory fetch)
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(0x80000000u), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
#else
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
#endif
}
// The non-temporal form (r01 measured it on the one-tile grouped kernel at a full mask: no difference; on the persistent
// kernel it pays where an expert has few rows, see gemm_fp8_persistent_kernel.hpp)
__device__ __forceinline__ void dma16_nt(uint32_t voff, v4i rsrc, uint32_t soff, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen nt lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_addr) : "memory");
}
// LDS-DMA, 4 B per lane from a per-lane 64-bit address (the strided scale gather)
__device__ __forceinline__ void dma4(const void *src, uint32_t lds_addr)
{
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %0, off"
                 :: "v"(src), "s"(lds_addr) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

}  // namespace dga
