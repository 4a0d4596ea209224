// Internal (non-ABI) declarations shared by the host translation units.
#pragma once
#include <hip/hip_runtime_api.h>

namespace dga {
int record_hip(hipError_t e);
// compiled fp8 kernel menu (dga_launch.hip)
int variant_count();
void variant_info(int i, int *bm, int *bn, int *wm, int *wn, int *lds);
}  // namespace dga
