// k_pass2.hip — pass 2 of the ZDW block encode on gfx950 (ConvertToZDW::writeBlockRows,
// cplusplus/ConvertToZDW.cpp:486-606).
//
// Input is the fixed-width record matrix rec[row][column] that pass 1 left in HBM (u32 dictionary
// table slot id per text field, u64 parsed value per numeric field), so this pass is pure
// fixed-stride integer work: stored value = dictionary offset | value - base, flag = stored value
// differs from the previous row's, row bytes = ceil(used/8) flag bytes followed by the low
// `size` bytes of every flagged value in column order.
//
// A row is handled by G = min(64, pow2 >= used columns) adjacent lanes (64/G rows per wave):
// flag bits come from one wave ballot, the byte position of each value from a G-lane prefix sum.
// Two launches: row lengths (then an exclusive scan gives each row's output offset), then emit.
#include "zdw_dev.h"
#include "zdw_kernels.h"

namespace zdw {

struct __attribute__((packed)) PackedW64 { uint64_t v; };
struct __attribute__((packed)) PackedW32 { uint32_t v; };
struct __attribute__((packed)) PackedW16 { uint16_t v; };

// what pass 1 left for (row, used column u): table slot id + 1 (text) or the parsed value (numeric)
__device__ __forceinline__ uint64_t raw_value(const Pass2Args &a, uint32_t row, uint32_t off, bool text)
{
    const uint8_t *p = a.rec + (uint64_t)row * a.rec_stride + off;
    return text ? (uint64_t)*reinterpret_cast<const uint32_t *>(p) : *reinterpret_cast<const uint64_t *>(p);
}
// the value the file stores for it: dictionary offset (ConvertToZDW.cpp:528-531) or value - base
// (:543-547, :564-566).  The mapping is injective per column (distinct strings have distinct offsets,
// v -> v - base is one-to-one, 0 <-> empty), so "differs from the previous row" can be decided on the
// raw values and the dictionary gather is only paid for values that are actually written.
__device__ __forceinline__ uint64_t stored_from_raw(const Pass2Args &a, uint64_t raw, uint32_t u, bool text)
{
    if (!raw) return 0ull;
    return text ? (uint64_t)a.slot_off[(uint32_t)raw - 1] : raw - a.used_base[u];
}
__device__ __forceinline__ uint64_t stored_value(const Pass2Args &a, uint32_t row, uint32_t u)
{
    const bool text = a.used_kind[u] == K_TEXT;
    return stored_from_raw(a, raw_value(a, row, a.used_recoff[u], text), u, text);
}

template <uint32_t G, bool EMIT>
__global__ __launch_bounds__(256) void k_pass2(const Pass2Args a)
{
    constexpr uint32_t ROWS_PER_WAVE = 64 / G;
    const uint32_t lane = lane_id();
    const uint32_t sub = lane & (G - 1), grp = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t row64 = wave * ROWS_PER_WAVE + grp;
    const bool row_ok = row64 < a.nrows;
    const uint32_t row = (uint32_t)row64;
    const uint32_t nflag = (a.nused + 7) >> 3;
    const uint64_t out_row = (EMIT && row_ok) ? a.row_off[row] : 0ull;
    uint32_t running = 0;      // value bytes of this row emitted so far (uniform inside the group)

    for (uint32_t u0 = 0; u0 < a.nused; u0 += G) {
        const uint32_t u = u0 + sub;
        const bool ok = row_ok && u < a.nused;
        uint64_t cur = 0;
        uint32_t sz = 0;
        bool flag = false, text = false;
        if (ok) {
            text = a.used_kind[u] == K_TEXT;
            const uint32_t off = a.used_recoff[u];
            cur = raw_value(a, row, off, text);
            sz = a.used_size[u];
            if (row) flag = cur != raw_value(a, row - 1, off, text);
            else flag = stored_from_raw(a, cur, u, text) != a.prev0[u];      // the halo row arrives as stored values
        }
        const uint32_t mine = flag ? sz : 0u;
        // inclusive prefix over the G lanes of the group
        uint32_t inc = mine;
#pragma unroll
        for (uint32_t d = 1; d < G; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, d, G);
            if (sub >= d) inc += t;
        }
        const uint32_t tot = (uint32_t)__shfl((int)inc, G - 1, G);
        if (EMIT) {
            const uint64_t bal = __ballot(flag);
            uint64_t bits = bal;
            if constexpr (G < 64) bits = (bal >> (grp * G)) & ((1ull << G) - 1ull);
            constexpr uint32_t FB = G >= 8 ? G / 8 : 1;                  // flag bytes this chunk can produce
            if (row_ok && sub < FB && u0 + 8 * sub < a.nused)
                a.out[out_row + (u0 >> 3) + sub] = (uint8_t)(bits >> (8 * sub));
            if (flag) {
                const uint64_t v = stored_from_raw(a, cur, u, text);
                // little-endian truncating store in at most three pieces (4 + 2 + 1 bytes; gfx950 stores unaligned)
                uint8_t *q = a.out + out_row + nflag + running + (inc - mine);
                if (sz == 8) { reinterpret_cast<PackedW64 *>(q)->v = v; }
                else {
                    uint64_t r = v;
                    if (sz & 4u) { reinterpret_cast<PackedW32 *>(q)->v = (uint32_t)r; q += 4; r >>= 32; }
                    if (sz & 2u) { reinterpret_cast<PackedW16 *>(q)->v = (uint16_t)r; q += 2; r >>= 16; }
                    if (sz & 1u) *q = (uint8_t)r;
                }
            }
        }
        running += tot;
    }
    if (!EMIT && row_ok && sub == 0) a.row_len[row] = nflag + running;
}

template <bool EMIT>
static void launch_pass2(const Pass2Args &a, hipStream_t s)
{
    if (!a.nrows) return;
    uint32_t g = 1;
    while (g < 64 && g < a.nused) g <<= 1;
    const uint64_t rows_per_block = (256 / 64) * (64 / g);
    const uint32_t grid = (uint32_t)((a.nrows + rows_per_block - 1) / rows_per_block);
    switch (g) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<1, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<2, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<4, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<8, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<16, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<32, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<64, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

// stored values of the block's last row (what the next shard of a row-sharded block compares its
// first row with); independent of the flags, so it needs no pass over the rows
__global__ __launch_bounds__(256) void k_last_row(const Pass2Args a)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < a.nused) a.last_stored[u] = stored_value(a, a.nrows - 1, u);
}

void launch_last_row(const Pass2Args &a, hipStream_t s)
{
    if (!a.nrows || !a.nused) return;
    hipLaunchKernelGGL(k_last_row, dim3((a.nused + 255) / 256), dim3(256), 0, s, a);
}

void launch_pass2_size(const Pass2Args &a, hipStream_t s) { launch_pass2<false>(a, s); }
void launch_pass2_emit(const Pass2Args &a, hipStream_t s) { launch_pass2<true>(a, s); }

}  // namespace zdw
