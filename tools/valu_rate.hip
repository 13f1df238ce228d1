// valu_rate.hip — wave64 issue cost of the integer instructions pass 1 is made of, on MI355X: cycles per wave-instruction per SIMD
// with 1, 2, 4, 8 waves per SIMD (independent chains, so only issue counts).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ __launch_bounds__(64) void k(uint32_t iters, uint32_t *out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint64_t b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    const uint32_t c = seed | 1u;
    for (uint32_t i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (OP == 0) { a0 += c; a1 += c; a2 += c; a3 += c; a4 += c; a5 += c; a6 += c; a7 += c; }                      // v_add_u32
            if (OP == 1) { a0 = (a0 & c) | a1; a1 = (a1 & c) | a2; a2 = (a2 & c) | a3; a3 = (a3 & c) | a4; a4 = (a4 & c) | a5; a5 = (a5 & c) | a6; a6 = (a6 & c) | a7; a7 = (a7 & c) | a0; }   // v_and_or / bitop3
            if (OP == 2) { a0 *= c; a1 *= c; a2 *= c; a3 *= c; a4 *= c; a5 *= c; a6 *= c; a7 *= c; }                      // v_mul_lo_u32
            if (OP == 3) { a0 = __umul24(a0, c) + 1; a1 = __umul24(a1, c) + 1; a2 = __umul24(a2, c) + 1; a3 = __umul24(a3, c) + 1; a4 = __umul24(a4, c) + 1; a5 = __umul24(a5, c) + 1; a6 = __umul24(a6, c) + 1; a7 = __umul24(a7, c) + 1; }   // v_mad_u32_u24
            if (OP == 4) { b0 += (uint64_t)a0 * a1; b1 += (uint64_t)a2 * a3; b2 += (uint64_t)a4 * a5; b3 += (uint64_t)a6 * a7; a0 ^= (uint32_t)b0; a2 ^= (uint32_t)b1; a4 ^= (uint32_t)b2; a6 ^= (uint32_t)b3; }   // v_mad_u64_u32 + v_xor
            if (OP == 5) { b0 = (b0 << 3) + b1; b1 = (b1 << 3) + b2; b2 = (b2 << 3) + b3; b3 = (b3 << 3) + b0; }       // v_lshl_add_u64
            if (OP == 6) { a0 = __builtin_amdgcn_alignbyte(a0, a1, c); a1 = __builtin_amdgcn_alignbyte(a1, a2, c); a2 = __builtin_amdgcn_alignbyte(a2, a3, c); a3 = __builtin_amdgcn_alignbyte(a3, a4, c); a4 = __builtin_amdgcn_alignbyte(a4, a5, c); a5 = __builtin_amdgcn_alignbyte(a5, a6, c); a6 = __builtin_amdgcn_alignbyte(a6, a7, c); a7 = __builtin_amdgcn_alignbyte(a7, a0, c); }
            if (OP == 7) { a0 = __builtin_popcount(a0) + a1; a1 = __builtin_popcount(a1) + a2; a2 = __builtin_popcount(a2) + a3; a3 = __builtin_popcount(a3) + a4; a4 = __builtin_popcount(a4) + a5; a5 = __builtin_popcount(a5) + a6; a6 = __builtin_popcount(a6) + a7; a7 = __builtin_popcount(a7) + a0; }   // v_bcnt
            if (OP == 8) { a0 = a0 > c ? a1 : a2; a1 = a1 > c ? a2 : a3; a2 = a2 > c ? a3 : a4; a3 = a3 > c ? a4 : a5; a4 = a4 > c ? a5 : a6; a5 = a5 > c ? a6 : a7; a6 = a6 > c ? a7 : a0; a7 = a7 > c ? a0 : a1; }   // v_cmp + v_cndmask
        }
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (uint32_t)(b0 ^ b1 ^ b2 ^ b3);
}

template <int OP>
int run(const char *name, int instr_per_round, uint32_t *out)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int wps : {1, 2, 4, 8}) {
        const uint32_t iters = 4096, grid = 256u * 4u * wps;
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, 16u, out, 3u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(64), 0, 0, iters, out, 3u);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        const double instr_per_simd = (double)wps * iters * 8.0 * instr_per_round;
        printf("%-28s waves/SIMD %d : %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz; %.3f ms)\n", name, wps, ms * 1e-3 * 2.4e9 / instr_per_simd, ms);
    }
    return 0;
}

int main()
{
    uint32_t *out = nullptr;
    CHECK(hipMalloc((void **)&out, 256 * 4 * 8 * 64 * 4));
    run<0>("v_add_u32", 8, out);
    run<1>("v_and_or / bitop3", 8, out);
    run<2>("v_mul_lo_u32", 8, out);
    run<3>("v_mad_u32_u24", 8, out);
    run<4>("v_mad_u64_u32 + v_xor", 8, out);
    run<5>("v_lshl_add_u64", 4, out);
    run<6>("v_alignbyte", 8, out);
    run<7>("v_bcnt + add", 8, out);
    run<8>("v_cmp + v_cndmask", 16, out);
    return 0;
}
