// probe_rate.hip — how many random 32-byte table probes (two `global_load_dwordx4 ... sc1`, the dictionary lookup of pass 1) does an
// MI355X serve per second, by table size, waves per CU and probes a lane keeps in flight?  (Is pass 1 near that rate, or waiting?)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int K, bool SC1>
__global__ __launch_bounds__(256) void k_probe(const uint64_t *__restrict__ slots, uint64_t slot_mask, uint32_t rounds, uint64_t *__restrict__ out)
{
    uint64_t x = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1), acc = 0;
    for (uint32_t r = 0; r < rounds; ++r) {
        u32x4 lo[K], hi[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
            const uint64_t *p = slots + 4 * (x & slot_mask);
            if (SC1) asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1" : "=&v"(lo[k]), "=&v"(hi[k]) : "v"(p) : "memory");
            else asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(lo[k]), "=&v"(hi[k]) : "v"(p) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < K; ++k) { asm volatile("" : "+v"(lo[k]), "+v"(hi[k])); acc += lo[k].x ^ hi[k].w; }
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

int main()
{
    uint64_t *out = nullptr;
    CHECK(hipMalloc((void **)&out, 64));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (uint64_t kb : {8ull, 16ull, 32ull, 64ull, 256ull, 1024ull, 2048ull, 4096ull, 32768ull, 1048576ull}) {
        void *t = nullptr;
        CHECK(hipMalloc(&t, kb << 10)); CHECK(hipMemset(t, 2, kb << 10));
        const uint64_t mask = ((kb << 10) / 32) - 1;
        for (int wpc : {24}) {
            for (int K : {1, -1}) {
                const bool sc1 = K > 0;
                const uint32_t grid = 256u * wpc / 4, rounds = 256;
                auto launch = [&]() {
                    if (sc1) hipLaunchKernelGGL((k_probe<1, true>), dim3(grid), dim3(256), 0, 0, (const uint64_t *)t, mask, rounds, out);
                    else hipLaunchKernelGGL((k_probe<1, false>), dim3(grid), dim3(256), 0, 0, (const uint64_t *)t, mask, rounds, out);
                };
                launch();
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(a));
                launch();
                CHECK(hipEventRecord(b));
                CHECK(hipEventSynchronize(b));
                float ms = 0;
                CHECK(hipEventElapsedTime(&ms, a, b));
                const double probes = (double)grid * 256 * rounds;
                printf("table %8llu KB  waves/CU %2d  %s : %7.2f G probes/s  (%.3f ms)\n", (unsigned long long)kb, wpc, sc1 ? "sc1  " : "plain", probes / ms / 1e6, ms);
            }
        }
        CHECK(hipFree(t));
    }
    return 0;
}
