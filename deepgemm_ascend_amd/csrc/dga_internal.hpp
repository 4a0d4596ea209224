// Internal (non-ABI) declarations shared by the host translation units.
#pragma once
#include <hip/hip_runtime_api.h>

#include "dga_hip.h"
#include <cstdint>

namespace dga {
int record_hip(hipError_t e);
// compiled fp8 kernel menu (dga_launch.hip)
int variant_count();
void variant_info(int i, int *bm, int *bn, int *wm, int *wn, int *lds);
int variant_stages(int i);
// dense 256x256 tilings: turn a small last partial wave into a K-split tail (dga_tiling.cpp)
void apply_tail_split(dga_tiling_t &t, uint32_t cus);
}  // namespace dga
