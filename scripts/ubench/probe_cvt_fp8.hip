// What v_cvt_pk_fp8_f32 does on gfx950 with out-of-range / NaN inputs (no clamp): decides which fix-ups the quantiser needs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
__global__ void k(const float *x, unsigned *out, int n)
{
    int i = threadIdx.x;
    if (i < n) out[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[i], 0.f, 0, false) & 0xFF;
}
int main()
{
    float h[16] = {NAN, -NAN, 1000.f, -1000.f, INFINITY, -INFINITY, 464.f, 465.f, 448.1f, 463.9f, 480.f, 1e-3f, -0.f, 0.0009765625f, 0.00146484375f, 3e38f};
    unsigned u; float nn; u = 0xFFC00000u; memcpy(&nn, &u, 4); h[1] = nn;
    float *d; unsigned *o; unsigned r[16];
    hipMalloc(&d, 64); hipMalloc(&o, 64);
    hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 16);
    hipMemcpy(r, o, 64, hipMemcpyDeviceToHost);
    for (int i = 0; i < 16; ++i) printf("%g -> 0x%02x\n", h[i], r[i]);
    return 0;
}
