// fetch_calib.hip — what does rocprofv3's FETCH_SIZE report on gfx950 for the two access patterns of pass 1?
//   k_stream   a coalesced streaming read, 16 bytes per lane (the TSV image)
//   k_probe    random 32-byte slots of a 1 GiB table, two `global_load_dwordx4 ... sc1` per lane (the dictionary table probe)
// Known byte counts are printed; run under `rocprofv3 --pmc FETCH_SIZE --output-format csv` and divide
// (tools/fetch_calib.sh does both and prints the factors).  MI355X_MICROARCH.md: FETCH_SIZE reads half the bytes
// of a wide coalesced stream; other widths are uncalibrated -- hence this tool.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stream(const uint4 *__restrict__ in, uint64_t n16, uint64_t *__restrict__ out)
{
    uint64_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint4 v = in[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

__global__ __launch_bounds__(256) void k_probe(const uint64_t *__restrict__ slots, uint64_t slot_mask, uint64_t probes_per_thread, uint64_t *__restrict__ out)
{
    uint64_t x = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1), acc = 0;
    for (uint64_t k = 0; k < probes_per_thread; ++k) {
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 29;
        const uint64_t *p = slots + 4 * (x & slot_mask);
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        u32x4 lo, hi;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
        acc += lo.x ^ hi.w;
    }
    if (acc == 0x123456789abcdefull) out[0] = acc;
}

int main()
{
    const uint64_t stream_bytes = 4ull << 30, table_bytes = 1ull << 30, nslots = table_bytes / 32;
    void *a = nullptr, *t = nullptr;
    uint64_t *out = nullptr;
    CHECK(hipMalloc(&a, stream_bytes)); CHECK(hipMalloc(&t, table_bytes)); CHECK(hipMalloc((void **)&out, 64));
    CHECK(hipMemset(a, 1, stream_bytes)); CHECK(hipMemset(t, 2, table_bytes));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_stream, dim3(256 * 8), dim3(256), 0, 0, (const uint4 *)a, stream_bytes / 16, out);
    CHECK(hipDeviceSynchronize());
    const uint64_t threads = 256ull * 8 * 256, per = 512;
    hipLaunchKernelGGL(k_probe, dim3(256 * 8), dim3(256), 0, 0, (const uint64_t *)t, nslots - 1, per, out);
    CHECK(hipDeviceSynchronize());
    printf("k_stream bytes %llu\nk_probe probes %llu bytes32 %llu\n", (unsigned long long)stream_bytes, (unsigned long long)(threads * per),
           (unsigned long long)(threads * per * 32));
    return 0;
}
