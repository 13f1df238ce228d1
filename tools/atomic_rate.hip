// atomic_rate.hip — how fast are fire-and-forget global atomic adds to scattered 32-bit counters (the question behind counting the next
// radix pass's per-chunk histogram inside the scatter of the pass before)?  n adds over a table of `bins` counters, addresses from a hash.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(64) void k(uint32_t *t, uint32_t bins, uint32_t n, uint32_t run)
{
    // one wave per 2048 elements, 32 steps (the shape of k_radix_scatter); `run` consecutive elements share a counter's neighbourhood
    const uint32_t beg = blockIdx.x * 2048;
    for (uint32_t k = 0; k < 32; ++k) {
        const uint32_t i = beg + 64 * k + threadIdx.x;
        if (i >= n) return;
        uint32_t h = (i / run) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        atomicAdd(&t[(h + (i % run)) % bins], 1u);
    }
}
int main()
{
    const uint32_t n = 5600000, bins = 256 * 2734;
    uint32_t *t = nullptr;
    CHECK(hipMalloc((void **)&t, (size_t)bins * 4));
    CHECK(hipMemset(t, 0, (size_t)bins * 4));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (uint32_t run : {1u, 8u, 64u}) {
        hipLaunchKernelGGL(k, dim3((n + 2047) / 2048), dim3(64), 0, 0, t, bins, n, run);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(a));
        for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k, dim3((n + 2047) / 2048), dim3(64), 0, 0, t, bins, n, run);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, a, b));
        printf("run %2u: %.1f us per launch of %u atomic adds over %u counters (%.1f G/s)\n", run, ms * 100.0, n, bins, n / (ms * 1e-4) * 1e-9);
    }
    return 0;
}
