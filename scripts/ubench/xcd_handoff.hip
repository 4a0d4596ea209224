// What does an in-launch split-K combine cost on MI355X when the two halves of a tile sit on CUs of ONE XCD (blocks b and
// b + 8 under the round-robin placement -- speed only, the protocol is placement-independent) against two different XCDs
// (b and b + 1)?  VERDICT r02 "next" item 7: measure the release / acquire pair there instead of assuming the ~1.7 us
// agent-scope figure.  Development aid, not part of the product.
//
// Every workgroup (256 threads, one per CU) produces a SLAB-byte fp32 slab (a decode tile's partial sums: 64 x 128 x 4 = 32 KB),
// publishes it, and the LAST ARRIVER of each pair (atomic ticket: no spinning, no co-residency assumption) sums both slabs and
// writes the bf16 tile.  Protocols:
//   0  no hand-off at all: every workgroup converts ITS OWN slab from registers (the floor)
//   1  plain stores -> every wave s_waitcnt vmcnt(0) -> __syncthreads -> lane 0: release fence (agent) + vmcnt(0) + ticket;
//      last arriver: acquire fence (agent) + vmcnt(0) + __syncthreads -> plain loads of the partner's slab
//   2  sc1 (write-through) stores -> vmcnt(0) -> __syncthreads -> ticket; last arriver: sc1 loads of the partner's slab
//      (MI355X_MICROARCH.md "Valid forms": every store and every load of the handed-off bytes sc1, no fence)
// Reported: launch time per protocol and pairing, and the difference to protocol 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int PROTO>
__global__ void __launch_bounds__(256) k(float *slabs, unsigned *tickets, unsigned short *out, int slab_floats, int pair_stride, int work)
{
    const int b = blockIdx.x, tid = threadIdx.x;
    // pair: blocks (p, p + pair_stride) inside groups of 2 * pair_stride blocks
    const int grp = b / (2 * pair_stride), in = b % (2 * pair_stride);
    const int half = in / pair_stride, pair = grp * pair_stride + in % pair_stride;
    const int partner = grp * 2 * pair_stride + (1 - half) * pair_stride + in % pair_stride;
    float *mine = slabs + (size_t)b * slab_floats;
    const float *theirs = slabs + (size_t)partner * slab_floats;
    // "compute": a dependent fma chain so that the workgroups do not all arrive in the same cycle
    v4f acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = v4f{(float)tid, 1.f, 2.f, (float)b};
    for (int it = 0; it < work + (b & 7) * 16; ++it)
        for (int i = 0; i < 8; ++i) acc[i] = acc[i] * 1.0001f + 0.5f;
    const int per_thread = slab_floats / 256 / 4;   // float4 per thread
    if (PROTO == 0) {
        for (int i = 0; i < per_thread; ++i) {
            const v4f v = acc[i & 7];
            const int at = (i * 256 + tid) * 4;
            out[(size_t)b * slab_floats / 2 + at / 2] = (unsigned short)(__builtin_bit_cast(unsigned, v.x + v.y) >> 16);
            out[(size_t)b * slab_floats / 2 + at / 2 + 1] = (unsigned short)(__builtin_bit_cast(unsigned, v.z + v.w) >> 16);
        }
        return;
    }
    for (int i = 0; i < per_thread; ++i) {
        v4f *dst = (v4f *)(mine + (i * 256 + tid) * 4);
        if (PROTO == 1) *dst = acc[i & 7];
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(acc[i & 7]) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ unsigned ticket;
    if (tid == 0) {
        if (PROTO == 1) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        ticket = __hip_atomic_fetch_add(&tickets[pair], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (PROTO == 1 && (ticket & 1)) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (!(ticket & 1)) return;           // first arriver: done (tickets count up launch after launch: parity tells)
    for (int i = 0; i < per_thread; ++i) {
        const int at = (i * 256 + tid) * 4;
        v4f o;
        if (PROTO == 1) o = *(const v4f *)(theirs + at);
        else asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(o) : "v"(theirs + at) : "memory");
        const v4f v = acc[i & 7] + o;
        out[(size_t)pair * slab_floats / 2 + at / 2] = (unsigned short)(__builtin_bit_cast(unsigned, v.x + v.y) >> 16);
        out[(size_t)pair * slab_floats / 2 + at / 2 + 1] = (unsigned short)(__builtin_bit_cast(unsigned, v.z + v.w) >> 16);
    }
}

int main(int argc, char **argv)
{
    const int slab_bytes = argc > 1 ? atoi(argv[1]) : 32768, work = argc > 2 ? atoi(argv[2]) : 2000, reps = 200;
    const int blocks = 256, slab_floats = slab_bytes / 4;
    float *slabs; unsigned *tickets; unsigned short *out;
    hipMalloc(&slabs, (size_t)blocks * slab_bytes); hipMalloc(&tickets, blocks * 4); hipMalloc(&out, (size_t)blocks * slab_bytes);
    hipMemset(tickets, 0, blocks * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("slab %d KB per workgroup, 256 workgroups x 256 threads, last arriver of each pair combines\n", slab_bytes / 1024);
    double base = 0;
    for (int proto = 0; proto < 3; ++proto)
        for (int stride : {8, 1}) {
            if (proto == 0 && stride == 1) continue;
            float ms = 0;
            for (int pass = 0; pass < 2; ++pass) {
                hipEventRecord(e0);
                for (int r = 0; r < reps; ++r) {
                    if (proto == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, slabs, tickets, out, slab_floats, stride, work);
                    else if (proto == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, slabs, tickets, out, slab_floats, stride, work);
                    else hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, slabs, tickets, out, slab_floats, stride, work);
                }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double us = ms * 1000 / reps;
            if (proto == 0) base = us;
            printf("protocol %d (%s), pairs (b, b+%d) = %s: %7.2f us per launch  (+%.2f us over the no-hand-off floor)\n", proto,
                   proto == 0 ? "no hand-off" : proto == 1 ? "plain stores + release / acquire fences" : "sc1 stores + sc1 loads, no fence",
                   stride, stride == 8 ? "one XCD" : "two XCDs", us, us - base);
        }
    return 0;
}
