// fp8 tile-kernel menu, part N: the bf16-exact policy's one-launch split-K for decode rows (gemm_fp8_bf16x_dsk_kernel.hpp;
// dispatchPolicyTag 7 with kernelSerial DGA_KERNEL_SPLITK_WORKGROUP and build DGA_BUILD_BX_DECODE): 64 x 128 tiles, two k groups per
// workgroup, `splits` workgroups per tile whose partial sums meet in the caller's workspace.
#include <atomic>
#include <cstdlib>
#include "dga_fp8_menu_impl.hpp"
#include "gemm_fp8_bf16x_dsk_kernel.hpp"
namespace dga {

static constexpr int kDskExtraHalfBlocks = 2;

// the splits a launch of `tiles` 64 x 128 tiles over kb k blocks runs with, given the tiling's splitkFactor: every workgroup resident
// at once (`cus`), every k slice at least two blocks, at most DskCfg::MAX_S workgroups per tile
int bx_dsk_splits(int64_t tiles, int kb, int want, int cus)
{
    if (tiles <= 0 || tiles > cus) return 0;
    int s = want > 0 ? want : 1;
    if (s > DskCfg::MAX_S) s = DskCfg::MAX_S;
    if (s > cus / tiles) s = static_cast<int>(cus / tiles);
    if (s > kb / 4) s = kb / 4;
    return s;      // 0: the problem is too short along K (fewer than four k blocks)
}

size_t bx_dsk_workspace_bytes(int64_t tiles, int splits)
{
    if (splits <= 1) return 0;
    return static_cast<size_t>(tiles) * (splits - 1) * (DskCfg::SLOT_FLOATS * 4 + 8) + 256;
}

template <bool KTAIL, bool IMG = true>
static int launch_bx_dsk_one(const GemmParams &p, const StreamKArgs &sk, unsigned grid, hipStream_t stream)
{
    auto kfn = gemm_fp8_bf16x_dsk_kernel<KTAIL, IMG>;
    constexpr int kLds = IMG ? DskCfg::IMG_LDS_BYTES : DskCfg::LDS_BYTES;
    static std::once_flag once[64];
    static hipError_t attr_err[64];
    int dev = 0;
    if (int rc = record_hip(hipGetDevice(&dev))) return rc;
    if (dev < 0 || dev >= 64) return DGA_E_HIP;
    std::call_once(once[dev], [&] {
        attr_err[dev] = hipFuncSetAttribute(reinterpret_cast<const void *>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    });
    if (int rc = record_hip(attr_err[dev])) return rc;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(DskCfg::NT), kLds, stream, p, sk);
    return record_hip(hipGetLastError());
}

// splits: the tiling's splitkFactor (clamped by bx_dsk_splits).  ws: the caller's workspace, bx_dsk_workspace_bytes() at least where the
// launch splits across workgroups.  DGA_E_TILING: not a launch this kernel takes (a layout it does not have, more tiles than CUs, fewer
// than four k blocks, co-residency not guaranteed, no or too small a workspace): the caller runs the tiling's tile kernel.
int launch_bf16x_dsk(const GemmParams &p, int splits, void *ws, size_t ws_bytes, hipStream_t stream)
{
    if (p.groups != 1 || p.masked_m || p.m_indices || p.row_index || p.splitk > 1 || p.tail_sub || p.stamps || p.launch_tiles) return DGA_E_TILING;
    const int cus = coresident_workgroups(stream);
    if (cus <= 0) return DGA_E_TILING;
    GemmParams q = p;
    q.tiles_m = (p.m + DskCfg::BM - 1) / DskCfg::BM;
    q.tiles_n = (p.n + DskCfg::BN - 1) / DskCfg::BN;
    const int64_t tiles = static_cast<int64_t>(q.tiles_m) * q.tiles_n;
    const int s = bx_dsk_splits(tiles, p.kb_n, splits, cus);
    if (s < 1) return DGA_E_TILING;
    q.splitk = s;
    StreamKArgs sk{};
    if (s > 1) {
        const size_t slots = static_cast<size_t>(tiles) * (s - 1);
        if (!ws || ws_bytes < slots * (DskCfg::SLOT_FLOATS * 4 + 8) || (reinterpret_cast<uintptr_t>(ws) & 15)) return DGA_E_TILING;
        sk.partials = static_cast<float *>(ws);
        sk.flags = reinterpret_cast<unsigned long long *>(sk.partials + slots * DskCfg::SLOT_FLOATS);
        // a flag is raised when it holds this launch's epoch -- 64 mixed bits no earlier launch used and stale workspace bytes will not
        // hold -- and its reader puts it back to 0, so a captured launch (replayed with these same arguments) needs no memset node
        static std::atomic<unsigned long long> launches{0};
        const unsigned long long e = launches.fetch_add(1) + 1;
        sk.epoch = (e * 0x9E3779B97F4A7C15ull) | 1ull;
    }
    // The adding workgroup's two slices are longer than the others' by what the hand-over costs (partial rows written through, the
    // flag, the read back: ~2.5 us, about `kDskExtraHalfBlocks` half k blocks of this loop), so that the partials are there when it has
    // multiplied its own: weights in half k blocks, wa = wo + extra (the kernel cuts K in proportion).
    if (s > 1) {
        int extra = kDskExtraHalfBlocks;
#ifdef DGA_DSK_KNOBS      // development builds: $DGA_DSK_KNOB (diagnostic bits: the kernel), $DGA_DSK_EXTRA (half blocks)
        if (const char *e = std::getenv("DGA_DSK_EXTRA")) extra = std::atoi(e);
#endif
        for (; extra > 0; --extra) {
            const int wo = (2 * p.kb_n - 2 * extra) / (2 * s), wa = wo + extra;
            if (wo < 4 || wa > 255 || wo > 255) continue;
            // every slice two k blocks at least, cut as the kernel cuts
            const int64_t W = 2 * wa + static_cast<int64_t>(2 * s - 2) * wo;
            auto cut = [&](int i) { return static_cast<int>((static_cast<int64_t>(p.kb_n) * (i <= 2 ? i * wa : 2 * wa + (i - 2) * wo)) / W); };
            bool ok = true;
            for (int i = 0; i < 2 * s && ok; ++i) ok = cut(i + 1) - cut(i) >= 2;
            if (ok) { q.tail_sub = wa | (wo << 8); break; }
        }
    }
#ifdef DGA_DSK_KNOBS
    if (const char *e = std::getenv("DGA_DSK_KNOB")) q.tail_begin = std::atoi(e);
#endif
    const unsigned grid = static_cast<unsigned>(tiles * s);
#ifdef DGA_DSK_KNOBS      // ($DGA_DSK_KNOB & 64: the build that converts A in every wave)
    if (q.tail_begin & 64)
        return (p.k % 128) ? launch_bx_dsk_one<true, false>(q, sk, grid, stream) : launch_bx_dsk_one<false, false>(q, sk, grid, stream);
#endif
    return (p.k % 128) ? launch_bx_dsk_one<true>(q, sk, grid, stream) : launch_bx_dsk_one<false>(q, sk, grid, stream);
}

}  // namespace dga
