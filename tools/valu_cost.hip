// valu_cost.hip -- cycles per wave64 instruction on one gfx950 SIMD with 8 resident waves, for the
// integer instructions the classify kernel is made of (cost weights for instruction-count tuning).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ __launch_bounds__(512, 8) void k(uint32_t *out, int iters, uint32_t seed)
{
    uint32_t a = threadIdx.x * 2654435761u + seed, b = a ^ 0x9E3779B9u, c = a + 77, d = b + 99;
    uint64_t q = ((uint64_t)a << 32) | b, r = ((uint64_t)c << 32) | d;
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 1) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 2) { REP64(asm volatile("v_lshlrev_b64 %0, 3, %0\n v_lshlrev_b64 %1, 5, %1" : "+v"(q), "+v"(r));) }
        if (OP == 3) { REP64(asm volatile("v_min_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 row_shr:2 row_mask:0xf bank_mask:0xf" : "+v"(a), "+v"(c));) }
        if (OP == 4) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "vcc");) }
        if (OP == 5) { REP64(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 6) { REP64(asm volatile("v_bfrev_b32 %0, %0\n v_bfrev_b32 %1, %1" : "+v"(a), "+v"(c));) }
        if (OP == 7) { REP64(asm volatile("v_cmp_lt_u64 vcc, %0, %1\n v_cmp_eq_u32 vcc, %2, %3" :: "v"(q), "v"(r), "v"(a), "v"(b) : "vcc");) }
        if (OP == 8) { REP64(asm volatile("v_lshrrev_b64 %0, %2, %0\n v_lshrrev_b64 %1, %2, %1" : "+v"(q), "+v"(r) : "v"(a & 7));) }
        if (OP == 9) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %2, %2, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 10) { REP64(asm volatile("v_and_or_b32 %0, %0, %1, %2\n v_bfe_u32 %3, %3, 3, 9" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));) }
        if (OP == 11) { REP64(asm volatile("v_lshl_add_u64 %0, %0, 4, %1\n v_lshl_add_u64 %1, %1, 4, %0" : "+v"(q), "+v"(r));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ (uint32_t)q ^ (uint32_t)(q >> 32) ^ (uint32_t)r;
}

template <int OP>
static void run(const char *name, uint32_t *out, int cus)
{
    const int iters = 200, grid = cus * 4; // 4 blocks x 8 waves = 32 waves per CU = 8 per SIMD
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(512), 0, 0, out, iters, 1u);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(512), 0, 0, out, iters, 2u);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = 8.0 * iters * 128; // 8 waves x iters x 64 reps x 2 instr
    printf("%-28s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (x clock GHz = cycles)\n", name, ms, ms * 1e6 / instr_per_simd);
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    uint32_t *out; CHECK(hipMalloc(&out, (size_t)p.multiProcessorCount * 4 * 512 * 4));
    printf("clock %d MHz\n", p.clockRate / 1000);
    run<0>("v_xor_b32", out, p.multiProcessorCount);
    run<5>("v_add_u32", out, p.multiProcessorCount);
    run<1>("v_mul_lo_u32", out, p.multiProcessorCount);
    run<9>("v_mul_hi_u32", out, p.multiProcessorCount);
    run<2>("v_lshlrev_b64 (imm)", out, p.multiProcessorCount);
    run<8>("v_lshrrev_b64 (vgpr)", out, p.multiProcessorCount);
    run<11>("v_lshl_add_u64", out, p.multiProcessorCount);
    run<3>("v_min_u32_dpp", out, p.multiProcessorCount);
    run<4>("v_cndmask_b32", out, p.multiProcessorCount);
    run<6>("v_bfrev_b32", out, p.multiProcessorCount);
    run<7>("v_cmp u64 / u32", out, p.multiProcessorCount);
    run<10>("v_and_or_b32 / v_bfe_u32", out, p.multiProcessorCount);
    return 0;
}
