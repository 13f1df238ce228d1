// k_scan.hip — exclusive prefix sums over u32 arrays (row starts per span, row lengths,
// radix histograms, dictionary string lengths).  HBM-bound: 4 B read + 4/8 B written per element,
// three launches: per-chunk sums, one-workgroup scan of the chunk sums, per-chunk scan.
#include "zdw_dev.h"
#include "zdw_kernels.h"

namespace zdw {

constexpr uint32_t SCAN_THREADS = 256;
constexpr uint32_t SCAN_ITEMS = 8;                           // per thread
constexpr uint32_t SCAN_CHUNK = SCAN_THREADS * SCAN_ITEMS;   // 2048 elements per workgroup

__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v, uint32_t lane)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, d, 64);
        uint32_t hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), d, 64);
        if (lane >= (uint32_t)d) v += ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// exclusive scan of one value per thread across a 256-thread workgroup; returns the exclusive
// prefix, *total = sum over the workgroup
__device__ inline uint64_t block_excl_scan(uint64_t v, uint64_t *total, uint64_t *lds /* >= 5 */)
{
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t inc = wave_incl_scan_u64(v, lane);
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    uint64_t base = 0, tot = 0;
#pragma unroll
    for (uint32_t i = 0; i < SCAN_THREADS / 64; ++i) {
        const uint64_t t = lds[i];
        if (i < wv) base += t;
        tot += t;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(SCAN_THREADS) void k_scan_chunk_sums(const uint32_t *__restrict__ in, uint64_t n,
                                                                 uint64_t *__restrict__ sums)
{
    __shared__ uint64_t lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) v += in[base + i];
    uint64_t tot;
    block_excl_scan(v, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// one workgroup: in-place exclusive scan of the chunk sums; writes the grand total
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_sums(uint64_t *sums, uint64_t nchunks, uint64_t *total)
{
    __shared__ uint64_t lds[8];
    uint64_t carry = 0;
    for (uint64_t b = 0; b < nchunks; b += SCAN_THREADS) {
        const uint64_t i = b + threadIdx.x;
        const uint64_t v = i < nchunks ? sums[i] : 0;
        uint64_t tot;
        const uint64_t ex = block_excl_scan(v, &tot, lds);
        if (i < nchunks) sums[i] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0 && total) *total = carry;
}

template <typename TOut>
__global__ __launch_bounds__(SCAN_THREADS) void k_scan_apply(const uint32_t *__restrict__ in, TOut *__restrict__ out,
                                                             uint64_t n, const uint64_t *__restrict__ sums)
{
    __shared__ uint64_t lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint32_t x[SCAN_ITEMS];
    uint64_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i) {
        x[i] = base + i < n ? in[base + i] : 0u;
        v += x[i];
    }
    uint64_t tot;
    uint64_t run = block_excl_scan(v, &tot, lds) + sums[blockIdx.x];
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) out[base + i] = (TOut)run;
        run += x[i];
    }
    // out[n] = total is written by the caller-visible convention: element n receives the grand sum
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) out[n] = (TOut)(sums[blockIdx.x] + tot);
}

size_t scan_tmp_bytes(uint64_t n) { return ((n + SCAN_CHUNK - 1) / SCAN_CHUNK + 1) * sizeof(uint64_t); }

template <typename TOut>
static void scan_impl(const uint32_t *in, TOut *out, uint64_t n, void *tmp, uint64_t *total, hipStream_t s)
{
    // out has n + 1 elements: out[n] = sum
    if (n == 0) {
        (void)hipMemsetAsync(out, 0, sizeof(TOut), s);
        if (total) (void)hipMemsetAsync(total, 0, sizeof(uint64_t), s);
        return;
    }
    const uint64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint64_t *sums = static_cast<uint64_t *>(tmp);
    hipLaunchKernelGGL(k_scan_chunk_sums, dim3((uint32_t)nchunks), dim3(SCAN_THREADS), 0, s, in, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, s, sums, nchunks, total);
    hipLaunchKernelGGL(HIP_KERNEL_NAME(k_scan_apply<TOut>), dim3((uint32_t)nchunks), dim3(SCAN_THREADS), 0, s, in, out, n, sums);
}

// Two exclusive scans in one: in[i] holds two flags (bit 0, bit 1); out0 / out1 (n + 1 elements each) = how many elements before i have
// bit 0 / bit 1 set.  The two counts travel as the halves of one 64-bit sum (n < 2^32), so the three launches serve both.
__global__ __launch_bounds__(SCAN_THREADS) void k_scan2_chunk_sums(const uint32_t *__restrict__ in, uint64_t n, uint64_t *__restrict__ sums)
{
    __shared__ uint64_t lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i)
        if (base + i < n) { const uint32_t x = in[base + i]; v += (uint64_t)(x & 1u) | ((uint64_t)((x >> 1) & 1u) << 32); }
    uint64_t tot;
    block_excl_scan(v, &tot, lds);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}
__global__ __launch_bounds__(SCAN_THREADS) void k_scan2_apply(const uint32_t *__restrict__ in, uint32_t *__restrict__ out0, uint32_t *__restrict__ out1,
                                                              uint64_t n, const uint64_t *__restrict__ sums)
{
    __shared__ uint64_t lds[8];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_CHUNK + (uint64_t)threadIdx.x * SCAN_ITEMS;
    uint64_t x[SCAN_ITEMS];
    uint64_t v = 0;
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i) {
        const uint32_t f = base + i < n ? in[base + i] : 0u;
        x[i] = (uint64_t)(f & 1u) | ((uint64_t)((f >> 1) & 1u) << 32);
        v += x[i];
    }
    uint64_t tot;
    uint64_t run = block_excl_scan(v, &tot, lds) + sums[blockIdx.x];
#pragma unroll
    for (uint32_t i = 0; i < SCAN_ITEMS; ++i) {
        if (base + i < n) { out0[base + i] = (uint32_t)run; out1[base + i] = (uint32_t)(run >> 32); }
        run += x[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) {
        const uint64_t all = sums[blockIdx.x] + tot;
        out0[n] = (uint32_t)all; out1[n] = (uint32_t)(all >> 32);
    }
}
void launch_scan_two_flags(const uint32_t *in, uint32_t *out0, uint32_t *out1, uint64_t n, void *tmp, hipStream_t s)
{
    if (n == 0) { (void)hipMemsetAsync(out0, 0, 4, s); (void)hipMemsetAsync(out1, 0, 4, s); return; }
    const uint64_t nchunks = (n + SCAN_CHUNK - 1) / SCAN_CHUNK;
    uint64_t *sums = static_cast<uint64_t *>(tmp);
    hipLaunchKernelGGL(k_scan2_chunk_sums, dim3((uint32_t)nchunks), dim3(SCAN_THREADS), 0, s, in, n, sums);
    hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(SCAN_THREADS), 0, s, sums, nchunks, (uint64_t *)nullptr);
    hipLaunchKernelGGL(k_scan2_apply, dim3((uint32_t)nchunks), dim3(SCAN_THREADS), 0, s, in, out0, out1, n, sums);
}

void launch_scan_u32_to_u32(const uint32_t *in, uint32_t *out, uint64_t n, void *tmp, uint64_t *total, hipStream_t s)
{
    scan_impl<uint32_t>(in, out, n, tmp, total, s);
}
void launch_scan_u32_to_u64(const uint32_t *in, uint64_t *out, uint64_t n, void *tmp, uint64_t *total, hipStream_t s)
{
    scan_impl<uint64_t>(in, out, n, tmp, total, s);
}

}  // namespace zdw
