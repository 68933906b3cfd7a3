// gather_shape.hip -- how does the SHAPE of a wave's 16-byte gather change the rate at which MI355X
// delivers distinct random 128-byte lines out of a 16 GiB table?  The classify kernel's header loads
// have runs of ~7-8 consecutive lanes on the same address (k-mers that share a minimizer), i.e. one
// line is asked for by two or three 4-lane quads of the same instruction.
//   shape 0: 64 lanes, 64 distinct lines                       (the plain gather ceiling)
//   shape 1: runs of R lanes on ONE address, all lanes issue   (what the kernel does)
//   shape 2: runs of R lanes, only the first lane of a run issues (exec-masked), 64/R lines
//   shape 3: runs of R lanes, every lane another cell of the same line
//   shape 4: runs of R lanes on one address, runs start at a random lane offset (not quad aligned)
//   shape 5: like 1, a random cell of the line instead of cell 0
// argv: [log2 cells] [fill: 0 = memset, 1 = hashed words] [MiB allocated (and kept) before the table] [MiB allocated before and freed]
// build: hipcc -O3 --offload-arch=gfx950 -o kmer_id_amd/bin/gather_shape tools/gather_shape.hip
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

template <int SHAPE, int R, int NT, int INFL = 4>
__global__ __launch_bounds__(256) void gather(const u4 *t, uint32_t line_mask, uint32_t rounds, uint32_t *sink)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (blockIdx.x * (uint64_t)blockDim.x + threadIdx.x) >> 6;
    uint32_t acc = 0;
    uint64_t ctr = wave * 0x9E3779B97F4A7C15ULL + 12345;
    for (uint32_t r = 0; r < rounds; r++) {
        u4 a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
        for (int u = 0; u < INFL; u++) {
            ctr += 0xD1B54A32D192ED03ULL;
            uint32_t shift = 0;
            if (SHAPE == 4) shift = (uint32_t)(ctr >> 40) % (uint32_t)R;
            const uint32_t run = SHAPE == 0 ? lane : (lane + shift) / (uint32_t)R;
            const uint32_t line = (uint32_t)fmix64(ctr ^ ((uint64_t)run << 48)) & line_mask;
            const u4 *p = t + (uint64_t)line * 8u + (SHAPE == 3 ? (lane & 7u) : SHAPE == 5 ? (uint32_t)(ctr >> 33) & 7u : 0u);
            // (the lane mask is applied INSIDE the statement: a load in a branch of its own gets its destination
            //  registers copied by the compiler before the data has landed)
            const uint64_t mask = SHAPE != 2 ? ~0ull : __ballot((lane % (uint32_t)R) == 0);
            uint64_t save;
            if (NT)
                asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_load_dwordx4 %0, %2, off nt\n\ts_mov_b64 exec, %1"
                             : "+v"(a[u]), "=&s"(save) : "v"(p), "s"(mask) : "memory");
            else
                asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %3\n\tglobal_load_dwordx4 %0, %2, off\n\ts_mov_b64 exec, %1"
                             : "+v"(a[u]), "=&s"(save) : "v"(p), "s"(mask) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : : "memory");
#pragma unroll
        for (int u = 0; u < 4; u++) acc ^= a[u].x ^ a[u].z;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <int SHAPE, int R, int NT, int INFL = 4>
static void run(const u4 *t, uint32_t line_mask, int grid, uint32_t *sink, const char *name)
{
    const uint64_t waves = (uint64_t)grid * 4;
    const uint32_t lines_per_instr = SHAPE == 0 ? 64 : (SHAPE == 4 ? (64 + R - 1) / R + 1 : (64 + R - 1) / R);
    uint32_t rounds = (uint32_t)(((uint64_t)1 << 27) / (waves * INFL * lines_per_instr));
    if (rounds < 4) rounds = 4;
    if (getenv("GS_MULT")) rounds *= (uint32_t)atoi(getenv("GS_MULT")); // longer kernels: is the rate of a 3 ms kernel sustained?
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather<SHAPE, R, NT, INFL>), dim3(grid), dim3(256), 0, 0, t, line_mask, rounds, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((gather<SHAPE, R, NT, INFL>), dim3(grid), dim3(256), 0, 0, t, line_mask, rounds, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 3;
    const double instrs = (double)rounds * waves * INFL;
    printf("%-44s %8.3f ms  %7.2f G lines/s  %7.2f G wave-loads/s\n", name, ms, instrs * lines_per_instr / (ms * 1e-3) / 1e9,
           instrs / (ms * 1e-3) / 1e9);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

__global__ void fill_hashed(u4 *t, size_t cells)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < cells; i += (size_t)gridDim.x * blockDim.x) {
        const uint64_t a = fmix64(i * 2 + 1), b = fmix64(i * 2 + 2);
        t[i] = u4{(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
    }
}

int main(int argc, char **argv)
{
    const int log2_cells = argc > 1 ? atoi(argv[1]) : 30;
    const int fill = argc > 2 ? atoi(argv[2]) : 0;
    const size_t keep_mib = argc > 3 ? (size_t)atol(argv[3]) : 0, junk_mib = argc > 4 ? (size_t)atol(argv[4]) : 0;
    const size_t cells = (size_t)1 << log2_cells;
    u4 *t; uint32_t *sink;
    void *keep = nullptr, *junk[64];
    if (keep_mib) CHECK(hipMalloc(&keep, keep_mib << 20));
    int nj = 0;
    for (size_t left = junk_mib; left > 0 && nj < 64; nj++) { // pieces of odd sizes, freed again: what a database build leaves behind
        const size_t piece = left > 700 ? 700 - 37 * (size_t)(nj % 5) : left;
        CHECK(hipMalloc(&junk[nj], piece << 20));
        CHECK(hipMemset(junk[nj], 0, piece << 20));
        left -= left > piece ? piece : left;
    }
    for (int j = 0; j < nj; j += 2) CHECK(hipFree(junk[j]));
    CHECK(hipMalloc(&t, cells * 16));
    for (int j = 1; j < nj; j += 2) CHECK(hipFree(junk[j]));
    if (fill) { hipLaunchKernelGGL(fill_hashed, dim3(4096), dim3(256), 0, 0, t, cells); CHECK(hipDeviceSynchronize()); }
    else CHECK(hipMemset(t, 1, cells * 16));
    printf("fill %d, %zu MiB kept and %zu MiB freed before the table, table at %p\n", fill, keep_mib, junk_mib, (void *)t);
    CHECK(hipMalloc(&sink, 64));
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int grid = p.multiProcessorCount * 8; // 8 x 256 threads per CU = 8 waves per SIMD, like the classify kernel
    const uint32_t line_mask = (uint32_t)((cells >> 3) - 1);
    printf("table %zu MiB, %d CUs, 4 loads in flight per lane, 8 waves per SIMD\n", cells * 16 >> 20, p.multiProcessorCount);
    run<0, 1, 0>(t, line_mask, grid, sink, "64 distinct lines per load");
    run<0, 1, 1>(t, line_mask, grid, sink, "64 distinct lines per load, nt");
    run<1, 4, 0>(t, line_mask, grid, sink, "runs of 4 (quad aligned), all lanes issue");
    run<1, 8, 0>(t, line_mask, grid, sink, "runs of 8 (2 quads), all lanes issue");
    run<1, 8, 1>(t, line_mask, grid, sink, "runs of 8 (2 quads), all lanes issue, nt");
    run<3, 8, 0>(t, line_mask, grid, sink, "runs of 8, 8 cells of the line");
    run<1, 7, 0>(t, line_mask, grid, sink, "runs of 7 (unaligned), all lanes issue");
    run<4, 7, 0>(t, line_mask, grid, sink, "runs of 7, random phase, all lanes issue");
    run<1, 16, 0>(t, line_mask, grid, sink, "runs of 16 (4 quads), all lanes issue");
    run<5, 8, 0>(t, line_mask, grid, sink, "runs of 8, a random cell of the line");
    if (getenv("GS_MLP")) { // lines in flight per CU = 32 waves x loads in flight x lines per load
        run<1, 8, 0, 1>(t, line_mask, grid, sink, "runs of 8, 1 load in flight (256 lines/CU)");
        run<1, 8, 0, 2>(t, line_mask, grid, sink, "runs of 8, 2 loads in flight (512 lines/CU)");
        run<1, 8, 0, 3>(t, line_mask, grid, sink, "runs of 8, 3 loads in flight (768 lines/CU)");
        run<1, 8, 0, 4>(t, line_mask, grid, sink, "runs of 8, 4 loads in flight (1024 lines/CU)");
        run<0, 1, 0, 1>(t, line_mask, grid, sink, "64 lines per load, 1 load in flight (2048 lines/CU)");
        run<0, 1, 0, 2>(t, line_mask, grid, sink, "64 lines per load, 2 loads in flight (4096 lines/CU)");
    }
    if (argc > 2) return 0;
    fprintf(stderr, "leaders-only variants\n");
    run<2, 4, 0>(t, line_mask, grid, sink, "runs of 4, leaders only");
    run<2, 8, 0>(t, line_mask, grid, sink, "runs of 8, leaders only");
    run<2, 8, 1>(t, line_mask, grid, sink, "runs of 8, leaders only, nt");
    run<2, 7, 0>(t, line_mask, grid, sink, "runs of 7, leaders only");
    run<2, 16, 0>(t, line_mask, grid, sink, "runs of 16, leaders only");
    return 0;
}
