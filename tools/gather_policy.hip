// gather_policy.hip -- does a cache-policy modifier on the 16-byte gather change the rate at
// which MI355X serves random cells out of a 16 GiB table (TLB/HBM bound)?
//   policies: 0 plain, 1 nt, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt, 5 sc0
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

template <int POL>
__device__ __forceinline__ void issue(u4 &dst, const u4 *p)
{
    if (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory");
    if (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(dst) : "v"(p) : "memory");
    if (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory");
    if (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(dst) : "v"(p) : "memory");
    if (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(dst) : "v"(p) : "memory");
    if (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(dst) : "v"(p) : "memory");
}

template <int POL>
__global__ __launch_bounds__(256) void gather(const u4 *t, uint32_t mask, uint32_t rounds, uint32_t *sink)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t ctr = tid * 0x9E3779B97F4A7C15ULL + 12345;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        u4 a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            ctr += 0xD1B54A32D192ED03ULL;
            issue<POL>(a[u], t + ((uint32_t)fmix64(ctr) & mask));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= a[u].x ^ a[u].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int POL>
static void run(const u4 *t, uint32_t mask, int grid, uint32_t *sink, const char *name, size_t mib)
{
    const uint64_t lanes = (uint64_t)grid * 256;
    uint32_t rounds = (uint32_t)(((uint64_t)1 << 28) / (lanes * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather<POL>), dim3(grid), dim3(256), 0, 0, t, mask, rounds, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((gather<POL>), dim3(grid), dim3(256), 0, 0, t, mask, rounds, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    printf("%-8zu %-12s %8.3f ms %8.2f Gcells/s\n", mib, name, ms, (double)rounds * lanes * 4 / (ms * 1e-3) / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int max_log2 = 30;
    const int alloc = argc > 1 ? atoi(argv[1]) : 0; // 0 hipMalloc, 1 uncached, 2 finegrained
    size_t cells = (size_t)1 << max_log2;
    u4 *t; uint32_t *sink;
    if (alloc == 0) CHECK(hipMalloc(&t, cells * 16));
    else if (alloc == 1) CHECK(hipExtMallocWithFlags((void **)&t, cells * 16, hipDeviceMallocUncached));
    else CHECK(hipExtMallocWithFlags((void **)&t, cells * 16, hipDeviceMallocFinegrained));
    CHECK(hipMemset(t, 1, cells * 16));
    CHECK(hipMalloc(&sink, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * 8;
    printf("alloc mode %d\n", alloc);
    for (int rl : {24, 26, 30}) {
        const uint32_t mask = (uint32_t)(((size_t)1 << rl) - 1);
        size_t mib = ((size_t)16 << rl) >> 20;
        run<0>(t, mask, grid, sink, "plain", mib);
        run<1>(t, mask, grid, sink, "nt", mib);
        run<2>(t, mask, grid, sink, "sc1", mib);
        run<3>(t, mask, grid, sink, "sc0 sc1", mib);
        run<4>(t, mask, grid, sink, "sc0 sc1 nt", mib);
        run<5>(t, mask, grid, sink, "sc0", mib);
    }
    return 0;
}
