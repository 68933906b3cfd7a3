// gather_sweep.hip -- random-gather micro-benchmarks on MI355X that price the hash
// lookup: how many random cells per second the memory system delivers, by region
// size (L2 / Infinity Cache / HBM), bytes touched per cell and loads in flight.
//   build: hipcc -O3 --offload-arch=gfx950 -o kmer_id_amd/bin/gather_sweep tools/gather_sweep.hip
//   modes: 0 = one 16 B load per cell
//          1 = two 16 B loads, same 64 B sector        (cell, cell^1)
//          2 = two 16 B loads, other half of a 128 B line (cell, cell^4)
//          3 = two 16 B loads, adjacent 128 B line      (cell, cell^8)
//          4 = four lanes share one 64 B sector (coalesced quad)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t fmix64(uint64_t k)
{
    k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
    return k;
}

template <int INF, int MODE>
__global__ __launch_bounds__(256) void gather(const uint4 *__restrict__ t, uint32_t mask, uint32_t rounds, uint32_t *sink)
{
    const uint64_t tid = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x;
    uint64_t ctr = (MODE == 4 ? (tid >> 2) : tid) * 0x9E3779B97F4A7C15ULL + 12345;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        uint4 a[INF], b[INF];
#pragma unroll
        for (int u = 0; u < INF; u++) {
            ctr += 0xD1B54A32D192ED03ULL;
            uint32_t idx = (uint32_t)fmix64(ctr) & mask;
            if (MODE == 4) idx = (idx & ~3u) | (uint32_t)(tid & 3);
            a[u] = t[idx];
            if (MODE == 1) b[u] = t[idx ^ 1u];
            if (MODE == 2) b[u] = t[idx ^ 4u];
            if (MODE == 3) b[u] = t[idx ^ 8u];
        }
#pragma unroll
        for (int u = 0; u < INF; u++) {
            acc ^= a[u].x ^ a[u].z;
            if (MODE >= 1 && MODE <= 3) acc ^= b[u].y;
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int INF, int MODE>
static float run(const uint4 *t, uint32_t mask, uint32_t rounds, int grid, uint32_t *sink, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather<INF, MODE>), dim3(grid), dim3(256), 0, 0, t, mask, rounds, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((gather<INF, MODE>), dim3(grid), dim3(256), 0, 0, t, mask, rounds, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

template <int MODE>
static float dispatch(int inf, const uint4 *t, uint32_t mask, uint32_t rounds, int grid, uint32_t *sink, int iters)
{
    switch (inf) {
    case 1: return run<1, MODE>(t, mask, rounds, grid, sink, iters);
    case 2: return run<2, MODE>(t, mask, rounds, grid, sink, iters);
    case 4: return run<4, MODE>(t, mask, rounds, grid, sink, iters);
    default: return run<8, MODE>(t, mask, rounds, grid, sink, iters);
    }
}

int main(int argc, char **argv)
{
    int max_log2 = argc > 1 ? atoi(argv[1]) : 30; // cells of 16 B
    size_t cells = (size_t)1 << max_log2;
    uint4 *t; uint32_t *sink;
    CHECK(hipMalloc(&t, cells * 16));
    CHECK(hipMemset(t, 1, cells * 16));
    CHECK(hipMalloc(&sink, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s, %d CUs, table %zu MiB\n", p.name, cus, cells * 16 >> 20);
    printf("%-8s %-5s %-4s %-5s %10s %12s %10s\n", "regionMiB", "mode", "inf", "wg/CU", "ms", "Gcells/s", "GB/s(16B)");
    const int region_log2[] = {18, 21, 22, 23, 24, 25, 26, 28, 30};
    for (int rl : region_log2) {
        if (rl > max_log2) continue;
        const uint32_t mask = (uint32_t)(((size_t)1 << rl) - 1);
        for (int mode = 0; mode <= 4; mode++) {
            for (int wgcu : {2, 4, 8}) {
                for (int inf : {1, 4}) {
                    if (mode != 0 && (wgcu != 8 || inf != 4) && !(mode == 0)) { if (!(wgcu == 8 && inf == 1)) continue; }
                    const int grid = cus * wgcu;
                    const uint64_t lanes = (uint64_t)grid * 256;
                    uint64_t target = (uint64_t)1 << 28;
                    uint32_t rounds = (uint32_t)(target / (lanes * inf));
                    if (rounds < 1) rounds = 1;
                    float ms = 0;
                    switch (mode) {
                    case 0: ms = dispatch<0>(inf, t, mask, rounds, grid, sink, 3); break;
                    case 1: ms = dispatch<1>(inf, t, mask, rounds, grid, sink, 3); break;
                    case 2: ms = dispatch<2>(inf, t, mask, rounds, grid, sink, 3); break;
                    case 3: ms = dispatch<3>(inf, t, mask, rounds, grid, sink, 3); break;
                    default: ms = dispatch<4>(inf, t, mask, rounds, grid, sink, 3); break;
                    }
                    double cellsps = (double)rounds * lanes * inf / (mode == 4 ? 4 : 1) / (ms * 1e-3);
                    printf("%-8zu %-5d %-4d %-5d %10.3f %12.2f %10.1f\n", ((size_t)16 << rl) >> 20, mode, inf, wgcu, ms, cellsps / 1e9,
                           cellsps * 16 / 1e9);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
