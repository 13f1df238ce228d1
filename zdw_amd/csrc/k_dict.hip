// k_dict.hip — the block dictionary after pass 1: an exact string sort of the unique strings
// (uniq_pos / uniq_len by unique id, as pass 1 left them; ids with length 0 are holes) into strcmp
// order (unsigned bytes, shorter prefix first: cplusplus/dictionary.h:30-35), offset assignment and the
// dictionary blob (Dictionary::write, cplusplus/dictionary.cpp:76-111).
//
// The sort is an MSD-by-8-bytes refinement: round d sorts the still-tied strings by
// (tie group, big-endian bytes [8d, 8d+8)) with a stable LSD radix sort (8-bit digits; histogram,
// scan, wave-ballot ranked scatter), then splits groups where keys differ.  Strings leave the
// working set as soon as they are distinguished, so the later rounds touch only long shared
// prefixes.  All integer work; bound by HBM traffic on (key,value) pairs.
#include "zdw_dev.h"
#include "zdw_knobs.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"

#include <algorithm>
#include <array>
#include <functional>
#include <vector>
#include <cstdio>
#include <cstdlib>

namespace zdw {

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

// ------------------------------------------------------------------------------------------
// LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits.  One wave per 2048-element chunk.
// ------------------------------------------------------------------------------------------
constexpr uint32_t SORT_CHUNK = 2048;
constexpr uint32_t SMALL_RUN = 64;          // tied runs of up to this many strings are finished by one wave each (k_sort_finish_small)

uint32_t sort_blocks(uint64_t n) { return (uint32_t)((n + SORT_CHUNK - 1) / SORT_CHUNK); }

// The digit of an element comes from its key, or -- DSRC -- from dsrc[value] (a second, more significant sort
// key looked up through the value; the pairs themselves keep travelling).
template <bool DSRC>
__global__ __launch_bounds__(64) void k_radix_hist(const uint64_t *__restrict__ keys, const uint32_t *__restrict__ vals,
                                                    const uint32_t *__restrict__ dsrc, uint32_t n, uint32_t shift,
                                                    uint32_t *__restrict__ hist, uint32_t nblocks)
{
    __shared__ uint32_t h[256];
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < 256; i += 64) h[i] = 0;
    __builtin_amdgcn_wave_barrier();          // one wave per block: LDS operations execute in order
    const uint32_t beg = blockIdx.x * SORT_CHUNK;
    const uint32_t end = beg + SORT_CHUNK < n ? beg + SORT_CHUNK : n;
    constexpr uint32_t STEPS = SORT_CHUNK / 64;
    // all loads of the chunk in flight before the first count
    uint32_t dig[STEPS];
    if (DSRC) {
        uint32_t v[STEPS];
#pragma unroll
        for (uint32_t k = 0; k < STEPS; ++k) { const uint32_t i = beg + 64 * k + lane; v[k] = i < end ? vals[i] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < STEPS; ++k) { const uint32_t i = beg + 64 * k + lane; dig[k] = i < end ? (dsrc[v[k]] >> shift) & 255u : 0u; }
    } else {
        // the digit's byte of the key only
        const uint8_t *kb = reinterpret_cast<const uint8_t *>(keys) + (shift >> 3);
#pragma unroll
        for (uint32_t k = 0; k < STEPS; ++k) { const uint32_t i = beg + 64 * k + lane; dig[k] = i < end ? kb[8ull * i] : 0u; }
    }
#pragma unroll
    for (uint32_t k = 0; k < STEPS; ++k)
        if (beg + 64 * k + lane < end) atomicAdd(&h[dig[k]], 1u);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t d = lane; d < 256; d += 64) hist[d * nblocks + blockIdx.x] = h[d];
}

// DSRC: the digit comes from dsrc[value] instead of the key.  The chunk is handled in two halves of SORT_STEPS
// 64-element steps; all loads of a half are issued before its first step is ranked (a step's rank needs the
// running offsets of the step before, so the steps themselves are sequential).
constexpr uint32_t SORT_STEPS = 16;
static_assert(SORT_CHUNK % (64 * SORT_STEPS) == 0, "a chunk is a whole number of half-chunks");

template <bool DSRC>
__global__ __launch_bounds__(64) void k_radix_scatter(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                       const uint32_t *__restrict__ dsrc, uint64_t *__restrict__ keys_out,
                                                       uint32_t *__restrict__ vals_out, uint32_t n, uint32_t shift,
                                                       const uint32_t *__restrict__ rows, const uint32_t *__restrict__ row_tot,
                                                       uint32_t nblocks)
{
    // One wave per block: its LDS operations execute in program order, so the running offsets need no
    // workgroup barrier (a barrier would also wait for the scattered stores of the step before).
    __shared__ uint32_t offs[256];
    const uint32_t lane = threadIdx.x;
    {   // where digit d of this block goes = elements with smaller digits + elements of digit d in earlier blocks
        const uint32_t t0 = row_tot[4 * lane], t1 = row_tot[4 * lane + 1], t2 = row_tot[4 * lane + 2], t3 = row_tot[4 * lane + 3];
        uint32_t tot;
        const uint32_t before = wave_excl_sum(t0 + t1 + t2 + t3, lane, tot);
        offs[4 * lane] = before + rows[(4 * lane) * nblocks + blockIdx.x];
        offs[4 * lane + 1] = before + t0 + rows[(4 * lane + 1) * nblocks + blockIdx.x];
        offs[4 * lane + 2] = before + t0 + t1 + rows[(4 * lane + 2) * nblocks + blockIdx.x];
        offs[4 * lane + 3] = before + t0 + t1 + t2 + rows[(4 * lane + 3) * nblocks + blockIdx.x];
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t beg = blockIdx.x * SORT_CHUNK;
    const uint32_t end = beg + SORT_CHUNK < n ? beg + SORT_CHUNK : n;
    for (uint32_t hb = beg; hb < end; hb += 64 * SORT_STEPS) {
        uint64_t key[SORT_STEPS];
        uint32_t val[SORT_STEPS], dig[SORT_STEPS];
#pragma unroll
        for (uint32_t k = 0; k < SORT_STEPS; ++k) {
            const uint32_t i = hb + 64 * k + lane;
            key[k] = i < end ? keys_in[i] : 0ull;
            val[k] = i < end ? vals_in[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < SORT_STEPS; ++k) {
            const uint32_t i = hb + 64 * k + lane;
            if (DSRC) dig[k] = i < end ? (dsrc[val[k]] >> shift) & 255u : 0u;
            else dig[k] = (uint32_t)(key[k] >> shift) & 255u;
        }
#pragma unroll
        for (uint32_t k = 0; k < SORT_STEPS; ++k) {
            const bool valid = hb + 64 * k + lane < end;
            const uint32_t digit = dig[k];
            uint64_t mm = __ballot(valid);                 // lanes holding the same digit as this lane
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (digit >> bit) & 1u;
                const uint64_t vote = __ballot(one);
                mm &= one ? vote : ~vote;
            }
            uint32_t dst = 0;
            if (valid) dst = offs[digit] + popc64(mm & mask_lt((int)lane));
            __builtin_amdgcn_wave_barrier();
            if (valid && lane == (uint32_t)highest_bit(mm)) offs[digit] += popc64(mm);
            __builtin_amdgcn_wave_barrier();
            if (valid) { keys_out[dst] = key[k]; vals_out[dst] = val[k]; }
        }
    }
}

// The same pass with the chunk put in order in LDS first (keys as digits: no DSRC).  Where a chunk's 2048 elements spread over all 256
// digits, the kernel above writes every element on its own -- 8 + 4 bytes into a 32-byte sector of its own: 1.0-1.3 ms per pass over
// 67 M pairs against 0.33 ms where a chunk holds one or two digits.  Here an element first goes to its place within the chunk (the
// chunk's own count of every digit is the difference of two neighbours of the scanned histogram), then the chunk is written out in
// that order: the ~8 elements of a digit lie side by side in LDS and go to consecutive addresses.  Stable like the other (an
// element's place among its digit's is its rank in chunk order).
__global__ __launch_bounds__(64) void k_radix_scatter_local(const uint64_t *__restrict__ keys_in, const uint32_t *__restrict__ vals_in,
                                                            uint64_t *__restrict__ keys_out, uint32_t *__restrict__ vals_out, uint32_t n, uint32_t shift,
                                                            const uint32_t *__restrict__ rows, const uint32_t *__restrict__ row_tot, uint32_t nblocks)
{
    __shared__ uint64_t lkey[SORT_CHUNK];
    __shared__ uint32_t lval[SORT_CHUNK];
    __shared__ uint32_t lofs[256];            // where the digit's next element goes within the chunk
    __shared__ uint32_t gofs[256];            // the digit's first global place minus its first place within the chunk
    const uint32_t lane = threadIdx.x;
    {
        const bool last = blockIdx.x + 1 == nblocks;
        uint32_t t[4], r[4], c[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            const uint32_t d = 4 * lane + k;
            t[k] = row_tot[d];
            r[k] = rows[d * nblocks + blockIdx.x];
            c[k] = (last ? t[k] : rows[d * nblocks + blockIdx.x + 1]) - r[k];      // this chunk's elements of digit d
        }
        uint32_t tot, ctot;
        uint32_t before = wave_excl_sum(t[0] + t[1] + t[2] + t[3], lane, tot);      // elements of smaller digits, all chunks
        uint32_t cbefore = wave_excl_sum(c[0] + c[1] + c[2] + c[3], lane, ctot);    // ... of this chunk
#pragma unroll
        for (uint32_t k = 0; k < 4; ++k) {
            lofs[4 * lane + k] = cbefore;
            gofs[4 * lane + k] = before + r[k] - cbefore;
            before += t[k]; cbefore += c[k];
        }
    }
    __builtin_amdgcn_wave_barrier();          // (one wave per block: its LDS operations execute in order)
    const uint32_t beg = blockIdx.x * SORT_CHUNK;
    const uint32_t end = beg + SORT_CHUNK < n ? beg + SORT_CHUNK : n;
    for (uint32_t hb = beg; hb < end; hb += 64 * SORT_STEPS) {
        uint64_t key[SORT_STEPS];
        uint32_t val[SORT_STEPS];
#pragma unroll
        for (uint32_t k = 0; k < SORT_STEPS; ++k) {
            const uint32_t i = hb + 64 * k + lane;
            key[k] = i < end ? keys_in[i] : 0ull;
            val[k] = i < end ? vals_in[i] : 0u;
        }
#pragma unroll
        for (uint32_t k = 0; k < SORT_STEPS; ++k) {
            const bool valid = hb + 64 * k + lane < end;
            const uint32_t digit = (uint32_t)(key[k] >> shift) & 255u;
            uint64_t mm = __ballot(valid);                 // lanes holding the same digit as this lane
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) {
                const bool one = (digit >> bit) & 1u;
                const uint64_t vote = __ballot(one);
                mm &= one ? vote : ~vote;
            }
            uint32_t at = 0;
            if (valid) at = lofs[digit] + popc64(mm & mask_lt((int)lane));
            __builtin_amdgcn_wave_barrier();
            if (valid && lane == (uint32_t)highest_bit(mm)) lofs[digit] += popc64(mm);
            __builtin_amdgcn_wave_barrier();
            if (valid) { lkey[at] = key[k]; lval[at] = val[k]; }
        }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t cnt = end - beg;
#pragma unroll 8
    for (uint32_t p = lane; p < cnt; p += 64) {
        const uint64_t k = lkey[p];
        const uint32_t dst = gofs[(uint32_t)(k >> shift) & 255u] + p;
        keys_out[dst] = k; vals_out[dst] = lval[p];
    }
}

// one block per digit: exclusive scan, in place, of the digit's per-block counts; the digit's total goes to row_tot.
// A thread owns up to 16 consecutive counts, all requested before any is used (one latency per 4096 counts).
__global__ __launch_bounds__(256) void k_radix_rowscan(uint32_t *__restrict__ hist, uint32_t nblocks, uint32_t *__restrict__ row_tot)
{
    constexpr uint32_t K = 16;
    __shared__ uint32_t wsum[4];
    uint32_t *row = hist + (size_t)blockIdx.x * nblocks;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t carry = 0;
    for (uint32_t b = 0; b < nblocks; b += 256 * K) {
        const uint32_t i0 = b + threadIdx.x * K;
        uint32_t x[K], v = 0;
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) { x[k] = i0 + k < nblocks ? row[i0 + k] : 0u; }
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) v += x[k];
        uint32_t wtot;
        const uint32_t ex = wave_excl_sum(v, lane, wtot);
        if (lane == 0) wsum[wv] = wtot;
        __syncthreads();
        uint32_t run = carry + ex, all = 0;
#pragma unroll
        for (uint32_t w = 0; w < 4; ++w) { const uint32_t t = wsum[w]; if (w < wv) run += t; all += t; }
#pragma unroll
        for (uint32_t k = 0; k < K; ++k) { if (i0 + k < nblocks) row[i0 + k] = run; run += x[k]; }
        carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0) row_tot[blockIdx.x] = carry;
}

// sorts by bits [bit_lo, bit_hi) of the key (or of dsrc[value]); result lands in (*keys, *vals) (pointers are swapped as needed)
static void radix_sort_pairs(uint64_t **keys, uint32_t **vals, uint64_t **keys_alt, uint32_t **vals_alt, const uint32_t *dsrc, uint32_t n,
                             uint32_t bit_lo, uint32_t bit_hi, uint32_t *hist, void *scan_tmp, hipStream_t s)
{
    const uint32_t nb = sort_blocks(n);
    // (ZDW_SORT_SCATTER: 0 = every element straight to its place, 1 = chunks put in order in LDS first; tests run both)
    // (measured: 67 M pairs, a packed round's six passes 12.3 -> 10.0 ms; C5's 36 M ids per block 7.4 -> 6.2 ms; the benchmark's 5.6 M
    // strings 2.14 -> 2.36 ms -- too few chunks to keep the chip busy at six waves per CU)
    bool local = n >= (1u << 24);
    if (const char *e = test_hook("ZDW_SORT_SCATTER")) local = atoi(e) != 0;
    for (uint32_t shift = bit_lo; shift < bit_hi; shift += 8) {
        if (dsrc) hipLaunchKernelGGL(k_radix_hist<true>, dim3(nb), dim3(64), 0, s, *keys, *vals, dsrc, n, shift, hist, nb);
        else hipLaunchKernelGGL(k_radix_hist<false>, dim3(nb), dim3(64), 0, s, *keys, *vals, dsrc, n, shift, hist, nb);
        uint32_t *row_tot = hist + (size_t)256 * nb;
        hipLaunchKernelGGL(k_radix_rowscan, dim3(256), dim3(256), 0, s, hist, nb, row_tot);
        if (dsrc) hipLaunchKernelGGL(k_radix_scatter<true>, dim3(nb), dim3(64), 0, s, *keys, *vals, dsrc, *keys_alt, *vals_alt, n, shift, hist, row_tot, nb);
        else if (local) hipLaunchKernelGGL(k_radix_scatter_local, dim3(nb), dim3(64), 0, s, *keys, *vals, *keys_alt, *vals_alt, n, shift, hist, row_tot, nb);
        else hipLaunchKernelGGL(k_radix_scatter<false>, dim3(nb), dim3(64), 0, s, *keys, *vals, dsrc, *keys_alt, *vals_alt, n, shift, hist, row_tot, nb);
        uint64_t *tk = *keys; *keys = *keys_alt; *keys_alt = tk;
        uint32_t *tv = *vals; *vals = *vals_alt; *vals_alt = tv;
    }
}

// ------------------------------------------------------------------------------------------
// refinement rounds
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sort_init(uint32_t *order, uint32_t *active, uint32_t *group, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { order[i] = i; active[i] = i; group[i] = 0; }
}

// ---- the first round's key, packed -------------------------------------------------------------------------------------------
// Tables hold strings whose characters vary little at a given place (date-times: a digit, a colon; ids with a common prefix): when the
// byte at place i of (nearly) all strings lies in one aligned block of 2^w values, only its low w bits decide the order.  A dictionary
// holds the strings of all text columns, so this is looked at per FIRST BYTE, on a sample of the strings (plan_packed_round, on the
// host): for each first byte and each of the places 1..23 kept by id (Pass1Args.uniq_key) the narrowest block that holds 98 % of the
// sample's bytes.  A string CONFORMS when every one of its places lies in its first byte's blocks (exact, on the device, for every
// string: k_sort_keys_packed); the others -- names among 67 M date-times, a date-time cut short -- are EXCEPTIONS.  Then
//   * the packed round sorts all the list's ids by one key: 0 for a hole,  1 | first byte (8 bits) | the low w bits of places
//     1 .. nchars - 1 (P bits)  for a conforming string, 2^(P+9) for an exception -- 16 or 24 characters in one round instead of 8,
//     on ceil((P + 10) / 8) radix passes instead of 8 -- and leaves holes | conforming strings in order | exceptions.  (The reference's
//     movie_tickets table times 128: the date-times' 23 places take 33 bits -- one round of 6 passes instead of three rounds of 8
//     and their tie-group passes.)  Order-preserving: strings with different first bytes differ above bit P; two conforming strings
//     of one first byte that first differ at place i agree in that place's dropped bits and in every bit before.  What is still
//     tied among the conforming strings goes on in ordinary rounds behind the characters every first byte's key covered;
//   * the exceptions (few, or the plan is dropped) are sorted by ordinary rounds -- up to 1024 of them by one workgroup;
//   * the two sorted lists are merged: every exception finds its place among the conforming strings by bisection on whole strings.
// The same for ONE LONG RUN of strings a round leaves tied over their first p0 = 8 or 16 bytes (dict_sort: sort_any): the byte at
// place p0 takes the first byte's part, the key holds the places behind it.
struct PackedBucket {
    uint64_t w[2];          // 4 bits per place 1..23: its width
    uint64_t himask[3];     // per byte of the string's three words: the bits above the width (none for place 0 and places >= nchars)
    uint64_t hival[3];      // ... and what a conforming string holds there
    uint32_t nchars, pad;   // 8, 16 or 24; 0: no string of this first byte conforms
};
static_assert(sizeof(PackedBucket) == 72, "SORT_SPARE_WORDS counts on it");
struct PackedKeySpec {
    const PackedBucket *table;      // by first byte (nullptr: the round takes 8 bytes as they are)
    uint32_t rest_bits;             // P
    uint32_t p0;                    // the place of that "first byte": 0 for a whole dictionary; 8 or 16 for one long run of strings that
                                    // are tied over their first p0 bytes (the packed key then holds places p0 + 1 .. nchars - 1)
};
constexpr uint32_t SORT_SAMPLE = 4096;

// out[4 t ..]: the three words and the length of string floor(t n / S) -- of a list of ids when there is one
__global__ __launch_bounds__(256) void k_sort_sample(const uint64_t *__restrict__ uniq_key, const uint32_t *__restrict__ uniq_len, uint32_t n, uint32_t S,
                                                     uint64_t *__restrict__ out, const uint32_t *__restrict__ list)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= S) return;
    const uint64_t at = (uint64_t)t * n / S;
    const uint64_t u = list ? list[at] : at;
    const uint32_t len = uniq_len[u];
    out[4 * t] = len ? uniq_key[3 * u] : 0; out[4 * t + 1] = len ? uniq_key[3 * u + 1] : 0; out[4 * t + 2] = len ? uniq_key[3 * u + 2] : 0;
    out[4 * t + 3] = len;
}

__device__ __forceinline__ bool conforms(uint64_t k0, uint64_t k1, uint64_t k2, const PackedBucket &pb)
{
    return pb.nchars && (k0 & pb.himask[0]) == pb.hival[0] && (k1 & pb.himask[1]) == pb.hival[1] && (k2 & pb.himask[2]) == pb.hival[2];
}

// (k0, k1, k2: a conforming string's first 24 bytes as kept by id -- its first character is the low byte of k0)
__device__ __forceinline__ uint32_t byte_at(uint64_t k0, uint64_t k1, uint64_t k2, uint32_t i)
{
    const uint64_t word = i < 8 ? k0 : (i < 16 ? k1 : k2);
    return (uint32_t)(word >> (8 * (i & 7u))) & 255u;
}
__device__ __forceinline__ uint64_t packed_key(uint64_t k0, uint64_t k1, uint64_t k2, const PackedBucket &pb, uint32_t rest_bits, uint32_t p0)
{
    uint64_t rest = 0;
#pragma unroll
    for (uint32_t i = 1; i < 24; ++i) {
        if (i <= p0) continue;
        if (i >= pb.nchars) break;
        const uint32_t w = (uint32_t)(pb.w[i >> 4] >> (4 * (i & 15u))) & 15u;
        rest = (rest << w) | (byte_at(k0, k1, k2, i) & ((1u << w) - 1u));
    }
    return ((uint64_t)(256u + byte_at(k0, k1, k2, p0)) << rest_bits) | rest;
}

// The packed round's keys, for all ids in place (vals[u] = u): a hole's key is 0, a conforming string's its packed key (bit P + 8
// set), an exception's 2^(P + 9) -- so the round leaves holes | conforming strings in order | exceptions, and k_sort_ties calls
// neither two holes nor two exceptions tied.  counts[c] += the ids of class c (0 hole, 1 conforming, 2 exception: a workgroup
// walks its share of the ids and adds what it counted once -- three atomics per 256 ids on three words took 3 ms at 67 M ids).
// (order / active: the list's j-th id is order[active[j]]; nullptr: j itself)
__global__ __launch_bounds__(256) void k_sort_keys_packed(const uint64_t *__restrict__ uniq_key, const uint32_t *__restrict__ uniq_len, uint32_t n,
                                                          const PackedKeySpec ps, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                          uint32_t *__restrict__ counts, const uint32_t *__restrict__ order,
                                                          const uint32_t *__restrict__ active)
{
    __shared__ uint32_t sc[3];
    if (threadIdx.x < 3) sc[threadIdx.x] = 0;
    __syncthreads();
    uint32_t mine0 = 0, mine1 = 0, mine2 = 0;
    const uint64_t per_block = (((uint64_t)n + gridDim.x - 1) / gridDim.x + 255) & ~255ull;
    const uint64_t beg = per_block * blockIdx.x, end = beg + per_block < n ? beg + per_block : n;
    for (uint64_t base = beg; base < end; base += 256) {
        const uint64_t j = base + threadIdx.x;
        const uint64_t u = j < end ? (order ? order[active[j]] : j) : 0;
        const bool real = j < end && uniq_len[u] != 0;
        uint64_t k0 = 0, k1 = 0, k2 = 0;
        if (real) { k0 = uniq_key[3 * u]; k1 = uniq_key[3 * u + 1]; k2 = uniq_key[3 * u + 2]; }
        uint64_t key = 0;
        const uint64_t rm = __ballot(real);
        if (rm) {
            // one first byte in the whole wave (a column of date-times): its entry of the table through scalar loads, once, instead of
            // nine gathers of 64 addresses
            const uint32_t b = byte_at(k0, k1, k2, ps.p0);
            const uint32_t b0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)lane_value(b, lowest_bit(rm)));
            if (__ballot(real && b != b0) == 0) {
                const PackedBucket pb = ps.table[b0];
                if (real) key = conforms(k0, k1, k2, pb) ? packed_key(k0, k1, k2, pb, ps.rest_bits, ps.p0) : 1ull << (ps.rest_bits + 9);
            } else if (real) {
                const PackedBucket pb = ps.table[b];
                key = conforms(k0, k1, k2, pb) ? packed_key(k0, k1, k2, pb, ps.rest_bits, ps.p0) : 1ull << (ps.rest_bits + 9);
            }
        }
        if (j < end) {
            keys[j] = key; vals[j] = (uint32_t)j;
            const uint32_t cls = key == 0 ? 0u : (key >> (ps.rest_bits + 9) ? 2u : 1u);
            mine0 += cls == 0u; mine1 += cls == 1u; mine2 += cls == 2u;
        }
    }
    const uint32_t t0 = wave_incl_sum(mine0), t1 = wave_incl_sum(mine1), t2 = wave_incl_sum(mine2);
    if (lane_id() == 63) { if (t0) atomicAdd(&sc[0], t0); if (t1) atomicAdd(&sc[1], t1); if (t2) atomicAdd(&sc[2], t2); }
    __syncthreads();
    if (threadIdx.x < 3 && sc[threadIdx.x]) atomicAdd(counts + threadIdx.x, sc[threadIdx.x]);
}

// keys[j] = 8 bytes at depth d of the string now at position active[j]; vals[j] = j
__global__ __launch_bounds__(256) void k_sort_keys(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint8_t *__restrict__ extra,
                                                    const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len,
                                                    const uint32_t *__restrict__ order, const uint32_t *__restrict__ active,
                                                    uint32_t m, uint32_t depth, uint64_t *__restrict__ keys,
                                                    uint32_t *__restrict__ vals, const uint64_t *__restrict__ uniq_key)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t u = order ? order[active[j]] : j;          // first round: everything is in place (order = active = identity)
    const uint32_t len = uniq_len[u];
    if (uniq_key && depth < 3) {
        // the first 24 bytes were kept by id when the string entered the table: in the first round a streaming read
        // instead of a visit to the image per string
        keys[j] = len ? __builtin_bswap64(uniq_key[3ull * u + depth]) : (uint64_t)u;
        vals[j] = j;
        return;
    }
    // a hole (length 0) sorts before every string (first byte >= 1) and apart from every other hole
    const uint64_t pos = uniq_pos[u];
    const uint8_t *p = string_at(in, extra, pos);
    keys[j] = len ? sort_key8(p, len, depth, pos >= EXTRA_BASE ? p + len + 16 : in + in_bytes) : (uint64_t)u;      // imported strings carry slack
    vals[j] = j;
}

// tmp[j] = the unique that belongs at the j-th tied position after this round
__global__ __launch_bounds__(256) void k_sort_gather(const uint32_t *__restrict__ order, const uint32_t *__restrict__ active,
                                                      const uint32_t *__restrict__ vals, uint32_t m, uint32_t *__restrict__ tmp,
                                                      uint32_t *__restrict__ max_run)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) *max_run = 0;                                  // (k_sort_ties, the next launch, takes the maximum)
    if (j < m) tmp[j] = order ? order[active[vals[j]]] : vals[j];
}

// writes the round's result back and flags what is still tied:
//   bit 1 of still[j] ("head") = position j starts a run of two or more equal (group, key)
//   bit 0 of still[j]          = position j is tied with a neighbour (needs another round)
// The radix sort's key array is aligned with the result (the tie-group passes move the pairs along), so no
// string is read here.
__global__ __launch_bounds__(256) void k_sort_ties(uint32_t *__restrict__ order, const uint32_t *__restrict__ active,
                                                    const uint32_t *__restrict__ group, const uint32_t *__restrict__ tmp,
                                                    const uint64_t *__restrict__ sorted_keys, uint32_t m,
                                                    uint32_t *__restrict__ still, uint32_t *__restrict__ max_run, uint64_t untied_from)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    order[active[j]] = tmp[j];
    const uint64_t k = sorted_keys[j];
    const uint32_t g = group[j];
    // (a packed round, k_sort_keys_packed: holes share key 0, exceptions the key untied_from, and neither are tied with anybody)
    const bool may_tie = !untied_from || (k != 0 && k < untied_from);
    const bool tie_prev = may_tie && j > 0 && group[j - 1] == g && sorted_keys[j - 1] == k;
    const bool tie_next = may_tie && j + 1 < m && group[j + 1] == g && sorted_keys[j + 1] == k;
    // run labels must stay dense (1..#runs) because the next round radix-sorts them on as few bits
    // as the number of still-tied strings needs: count a head only for runs that stay in the working set
    still[j] = ((tie_prev || tie_next) ? 1u : 0u) | ((!tie_prev && tie_next) ? 2u : 0u);
    if (!tie_prev && tie_next) {
        // is the run this position starts longer than SMALL_RUN?  One look at the element SMALL_RUN places on (the run is contiguous).
        // The host asks how many such runs there are: none -- k_sort_finish_small ends the sort here; one -- it finishes all the others
        // and the next round sorts that one run alone (no tie-group passes)
        if (j + SMALL_RUN < m && group[j + SMALL_RUN] == g && sorted_keys[j + SMALL_RUN] == k) atomicAdd(max_run, 1u);      // (long runs are few: one count each)
    }
}

__global__ __launch_bounds__(256) void k_sort_compact(const uint32_t *__restrict__ active,
                                                       const uint32_t *__restrict__ still, const uint32_t *__restrict__ head_scan,
                                                       const uint32_t *__restrict__ still_scan, uint32_t m,
                                                       uint32_t *__restrict__ active_out, uint32_t *__restrict__ group_out,
                                                       uint32_t *__restrict__ run_start)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= m) return;
    const uint32_t f = still[j];
    if (!(f & 1u)) return;
    const uint32_t k = still_scan[j], head = (f >> 1) & 1u;
    active_out[k] = active[j];
    group_out[k] = head_scan[j] + head;         // run label: grows with position, equal inside a run
    if (head) run_start[head_scan[j]] = k;      // where run number label - 1 begins in the next round's working set
}

// The end of the sort when every tied run is small: one wave per run, a lane per string, every string compared with every other of
// its run (the strings' first 24 bytes from the words kept by id where there are any, what lies behind from the image); the lane's
// rank is where its string belongs.  The benchmark's table ends its second round with 1.7 M date-times tied in runs of up to 60 (one
// per minute): this kernel instead of ten more radix passes.
__global__ __launch_bounds__(256) void k_sort_finish_small(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint8_t *__restrict__ extra,
                                                            const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len,
                                                            const uint64_t *__restrict__ uniq_key, const uint32_t *__restrict__ order,
                                                            const uint32_t *__restrict__ active, const uint32_t *__restrict__ run_start,
                                                            uint32_t nruns, uint32_t m, uint32_t *__restrict__ tmp, uint32_t *__restrict__ err)
{
    __shared__ uint64_t K[4][3][64];
    __shared__ uint32_t U[4][64];
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t r = blockIdx.x * 4u + wv;
    if (r >= nruns) return;
    const uint32_t s0 = run_start[r], e0 = r + 1 < nruns ? run_start[r + 1] : m;
    const uint32_t len = e0 - s0;
    if (e0 > m || e0 <= s0) { if (lane == 0) err[0] = 1; return; }      // (cannot happen)
    if (len > SMALL_RUN) { if (lane == 0) { err[1] = s0; err[2] = e0; } return; }      // the one long run (the host has counted): where it lies, for the next round
    auto key_at = [&](uint32_t u, uint32_t ulen, uint32_t depth) -> uint64_t {
        if (uniq_key && depth < 3) return __builtin_bswap64(uniq_key[3ull * u + depth]);
        const uint64_t pos = uniq_pos[u];
        const uint8_t *p = string_at(in, extra, pos);
        return sort_key8(p, ulen, depth, pos >= EXTRA_BASE ? p + ulen + 16 : in + in_bytes);
    };
    uint32_t u = 0, ulen = 0;
    uint64_t k0 = 0, k1 = 0, k2 = 0;
    if (lane < len) {
        u = order[active[s0 + lane]];
        ulen = uniq_len[u];
        k0 = key_at(u, ulen, 0); k1 = key_at(u, ulen, 1); k2 = key_at(u, ulen, 2);
        K[wv][0][lane] = k0; K[wv][1][lane] = k1; K[wv][2][lane] = k2; U[wv][lane] = u;
    }
    __builtin_amdgcn_wave_barrier();          // (one wave per run: its LDS operations execute in order)
    uint32_t rank = 0;
    for (uint32_t j = 0; j < len; ++j) {
        const uint64_t o0 = K[wv][0][j], o1 = K[wv][1][j], o2 = K[wv][2][j];
        if (lane >= len || j == lane) continue;
        bool less;                                                 // the other string sorts before mine
        if (o0 != k0) less = o0 < k0;
        else if (o1 != k1) less = o1 < k1;
        else if (o2 != k2) less = o2 < k2;
        else {
            // equal in their first 24 bytes: both are longer; on through the image, 8 bytes at a time (zero padding: the shorter first)
            const uint32_t ou = U[wv][j], olen = uniq_len[ou];
            less = false;
            for (uint32_t d = 3; 8u * d < ulen || 8u * d < olen; ++d) {
                const uint64_t a = key_at(ou, olen, d), b = key_at(u, ulen, d);
                if (a != b) { less = a < b; break; }
            }
        }
        rank += less ? 1u : 0u;
    }
    if (lane < len) tmp[s0 + rank] = u;
}

// The same for runs of two or three strings (hundreds of thousands of them after a first round: a wave each would be a wave of two
// busy lanes): a THREAD per string, which walks its run and counts the members that sort before it.  Strings of the one long run
// leave at once.
__global__ __launch_bounds__(256) void k_sort_finish_tiny(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint8_t *__restrict__ extra,
                                                           const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len,
                                                           const uint64_t *__restrict__ uniq_key, const uint32_t *__restrict__ order,
                                                           const uint32_t *__restrict__ active, const uint32_t *__restrict__ group,
                                                           const uint32_t *__restrict__ run_start, uint32_t nruns, uint32_t m,
                                                           uint32_t *__restrict__ tmp, uint32_t *__restrict__ err)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const uint32_t r = group[k] - 1u;
    if (r >= nruns) { err[0] = 1; return; }                             // (cannot happen)
    const uint32_t s0 = run_start[r], e0 = r + 1 < nruns ? run_start[r + 1] : m;
    if (e0 > m || e0 <= s0 || k < s0 || k >= e0) { err[0] = 1; return; }      // (cannot happen)
    if (e0 - s0 > SMALL_RUN) { if (k == s0) { err[1] = s0; err[2] = e0; } return; }
    auto key_at = [&](uint32_t u, uint32_t ulen, uint32_t depth) -> uint64_t {
        if (uniq_key && depth < 3) return __builtin_bswap64(uniq_key[3ull * u + depth]);
        const uint64_t pos = uniq_pos[u];
        const uint8_t *p = string_at(in, extra, pos);
        return sort_key8(p, ulen, depth, pos >= EXTRA_BASE ? p + ulen + 16 : in + in_bytes);
    };
    const uint32_t u = order[active[k]], ulen = uniq_len[u];
    const uint64_t k0 = key_at(u, ulen, 0), k1 = key_at(u, ulen, 1), k2 = key_at(u, ulen, 2);
    uint32_t rank = 0;
    for (uint32_t i = s0; i < e0; ++i) {
        if (i == k) continue;
        const uint32_t ou = order[active[i]], olen = uniq_len[ou];
        const uint64_t o0 = key_at(ou, olen, 0);
        bool less;
        if (o0 != k0) less = o0 < k0;
        else {
            const uint64_t o1 = key_at(ou, olen, 1);
            if (o1 != k1) less = o1 < k1;
            else {
                const uint64_t o2 = key_at(ou, olen, 2);
                if (o2 != k2) less = o2 < k2;
                else {
                    less = false;
                    for (uint32_t d = 3; 8u * d < ulen || 8u * d < olen; ++d) {
                        const uint64_t a = key_at(ou, olen, d), b = key_at(u, ulen, d);
                        if (a != b) { less = a < b; break; }
                    }
                }
            }
        }
        rank += less ? 1u : 0u;
    }
    tmp[s0 + rank] = u;
}

// (skip: the run k_sort_finish_small left for the next round, [skip[0], skip[1]); nullptr: none)
__global__ __launch_bounds__(256) void k_sort_place(uint32_t *__restrict__ order, const uint32_t *__restrict__ active, const uint32_t *__restrict__ tmp, uint32_t m,
                                                    const uint32_t *__restrict__ skip)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    if (skip && k >= skip[0] && k < skip[1]) return;
    order[active[k]] = tmp[k];
}

// ---- the merge of the conforming strings' list and the exceptions' ----
// a < b as whole strings (both real): the words kept by id, then the image
__device__ inline bool string_less(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                                   const uint64_t *uniq_key, uint32_t a, uint32_t b)
{
    const uint32_t la = uniq_len[a], lb = uniq_len[b];
    for (uint32_t d = 0; d < 3; ++d) {
        const uint64_t ka = __builtin_bswap64(uniq_key[3ull * a + d]), kb = __builtin_bswap64(uniq_key[3ull * b + d]);
        if (ka != kb) return ka < kb;
    }
    const uint64_t pa = uniq_pos[a], pb = uniq_pos[b];
    const uint8_t *sa = string_at(in, extra, pa), *sb = string_at(in, extra, pb);
    for (uint32_t d = 3; 8u * d < la || 8u * d < lb; ++d) {
        const uint64_t ka = sort_key8(sa, la, d, pa >= EXTRA_BASE ? sa + la + 16 : in + in_bytes);
        const uint64_t kb = sort_key8(sb, lb, d, pb >= EXTRA_BASE ? sb + lb + 16 : in + in_bytes);
        if (ka != kb) return ka < kb;
    }
    return false;
}
// rank[j] = how many of the sorted strings A[0 .. na) sort before the j-th of the sorted strings B
__global__ __launch_bounds__(256) void k_merge_ranks(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint8_t *__restrict__ extra,
                                                     const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len,
                                                     const uint64_t *__restrict__ uniq_key, const uint32_t *__restrict__ A, uint32_t na,
                                                     const uint32_t *__restrict__ B, uint32_t nb, uint32_t *__restrict__ rank)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nb) return;
    const uint32_t u = B[j];
    uint32_t lo = 0, hi = na;
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (string_less(in, in_bytes, extra, uniq_pos, uniq_len, uniq_key, A[mid], u)) lo = mid + 1; else hi = mid;
    }
    rank[j] = lo;
}
// out[0 .. na + nb): the two lists as one.  B's j-th string lands at rank[j] + j; A's i-th behind the exceptions that sort before it
// (rank <= i; the ranks ascend: bisection).
__global__ __launch_bounds__(256) void k_merge_place(const uint32_t *__restrict__ A, uint32_t na, const uint32_t *__restrict__ B, uint32_t nb,
                                                     const uint32_t *__restrict__ rank, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= na + nb) return;
    if (i >= na) { const uint32_t j = i - na; out[rank[j] + j] = B[j]; return; }
    uint32_t lo = 0, hi = nb;                        // the first j with rank[j] > i
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (rank[mid] <= i) lo = mid + 1; else hi = mid;
    }
    out[i + lo] = A[i];
}
// A list of up to SORT_TINY strings (the exceptions beside a packed round; a dictionary of a few hundred strings): one workgroup,
// every string compared with every other as whole strings, its rank is where it goes -- one launch instead of the ~30 a round of
// radix passes takes whatever its size.  list[0 .. m) -> out[0 .. m) sorted (out may not alias list).
constexpr uint32_t SORT_TINY = 1024;
__global__ __launch_bounds__(256) void k_sort_tiny(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint8_t *__restrict__ extra,
                                                   const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len,
                                                   const uint64_t *__restrict__ uniq_key, const uint32_t *__restrict__ list, uint32_t m,
                                                   uint32_t *__restrict__ out)
{
    __shared__ uint32_t ids[SORT_TINY];
    for (uint32_t i = threadIdx.x; i < m; i += 256) ids[i] = list[i];
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += 256) {
        const uint32_t u = ids[i];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < m; ++j)
            if (j != i && string_less(in, in_bytes, extra, uniq_pos, uniq_len, uniq_key, ids[j], u)) ++rank;
        out[rank] = u;
    }
}
__global__ __launch_bounds__(256) void k_sort_init_range(uint32_t *active, uint32_t *group, uint32_t off, uint32_t m)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) { active[off + i] = off + i; group[off + i] = 0; }
}

// What the packed first round would look like, from a sample of the strings (SORT_SAMPLE x {three words, length}).  force (tests):
// a plan whenever one exists, however little it saves and however many strings it leaves out.
struct PackPlan { bool use = false; uint32_t P = 0, depth_after = 0, chars = 0; double exceptions = 0.0; std::vector<PackedBucket> table; };
// p0: the strings of the sample are tied over their first p0 bytes (a long run behind round p0 / 8): the byte at place p0 takes the
// first byte's part, the packed key holds the places behind it.
static void plan_packed_round(const uint64_t *sample, uint32_t S, bool force, uint32_t p0, PackPlan &pl)
{
    auto byte_of = [&](uint32_t t, uint32_t i) { return (uint32_t)(sample[4 * t + (i >> 3)] >> (8 * (i & 7u))) & 255u; };
    std::vector<std::vector<uint32_t>> by(256);
    uint32_t real = 0;
    for (uint32_t t = 0; t < S; ++t) if (sample[4 * t + 3]) { by[byte_of(t, p0)].push_back(t); ++real; }
    if (real < (force ? 2u : 256u)) return;
    std::vector<std::array<uint32_t, 24>> w(256);
    std::vector<std::array<uint32_t, 25>> W(256);
    std::vector<std::array<uint8_t, 24>> base(256);
    std::vector<uint32_t> judged;
    for (uint32_t b = 0; b < 256; ++b) {
        const uint32_t cnt = (uint32_t)by[b].size();
        if (cnt < (force ? 1u : 16u)) continue;                    // too few to judge: exceptions
        for (uint32_t i = 0; i <= p0; ++i) { W[b][i] = 0; w[b][i] = 0; base[b][i] = 0; }
        W[b][p0 + 1] = 0;
        const uint32_t need = cnt - cnt / 50;                      // 98 % of them
        for (uint32_t i = p0 + 1; i < 24; ++i) {
            uint32_t upto[257] = {0};                               // upto[v]: the strings whose byte here is below v
            for (uint32_t t : by[b]) ++upto[byte_of(t, i) + 1];
            for (uint32_t v = 0; v < 256; ++v) upto[v + 1] += upto[v];
            uint32_t wi = 8, bs = 0;
            for (uint32_t k = 0; k < 8 && wi == 8; ++k)            // the narrowest aligned block of 2^k values that holds them
                for (uint32_t lo = 0; lo < 256; lo += 1u << k)
                    if (upto[lo + (1u << k)] - upto[lo] >= need) { wi = k; bs = lo; break; }
            w[b][i] = wi; base[b][i] = (uint8_t)bs;
            W[b][i + 1] = W[b][i] + wi;
        }
        judged.push_back(b);
    }
    if (judged.empty()) return;
    std::sort(judged.begin(), judged.end(), [&](uint32_t x, uint32_t y) { return by[x].size() != by[y].size() ? by[x].size() > by[y].size() : x < y; });
    // P: what the first bytes that hold 98 % of the sample need for 24, 16 or 8 characters (the key has 63 - 9 bits for them)
    uint32_t nmajor = 0;
    for (uint64_t seen = 0; nmajor < judged.size() && seen * 50 < (uint64_t)real * 49; ++nmajor) seen += by[judged[nmajor]].size();
    uint32_t kmaj = 0, P = 0;
    for (uint32_t k : {24u, 16u, 8u}) {
        if (k <= p0) break;
        uint32_t need = 0;
        for (uint32_t i = 0; i < nmajor; ++i) need = std::max(need, W[judged[i]][k]);
        if (need + 9 <= 63) { kmaj = k; P = need; break; }
    }
    if (!kmaj) return;
    pl.table.assign(256, PackedBucket{});
    pl.depth_after = 3;
    for (uint32_t b : judged) {
        PackedBucket &pb = pl.table[b];
        pb.nchars = W[b][24] <= P ? 24u : (W[b][16] <= P && p0 < 16 ? 16u : (W[b][8] <= P && p0 < 8 ? 8u : 0u));
        for (uint32_t i = p0 + 1; i < pb.nchars; ++i) {
            pb.w[i >> 4] |= (uint64_t)w[b][i] << (4 * (i & 15u));
            const uint64_t hm = (0xFFu << w[b][i]) & 0xFFu;
            pb.himask[i >> 3] |= hm << (8 * (i & 7u));
            pb.hival[i >> 3] |= ((uint64_t)base[b][i] & hm) << (8 * (i & 7u));
        }
        if (pb.nchars) pl.depth_after = std::min(pl.depth_after, pb.nchars / 8);
    }
    uint32_t exc = 0;
    for (uint32_t t = 0; t < S; ++t) {
        if (!sample[4 * t + 3]) continue;
        const PackedBucket &pb = pl.table[byte_of(t, p0)];
        const bool ok = pb.nchars && (sample[4 * t] & pb.himask[0]) == pb.hival[0] && (sample[4 * t + 1] & pb.himask[1]) == pb.hival[1] &&
                        (sample[4 * t + 2] & pb.himask[2]) == pb.hival[2];
        if (!ok) ++exc;
    }
    pl.P = P; pl.chars = kmaj; pl.exceptions = (double)exc / (double)real;
    // worth it: few exceptions (they are sorted the ordinary way and then found their places one by one), and more than 8 characters
    // in the round or 6 passes at most
    pl.use = force ? exc < real : (exc * 32ull <= real && (kmaj - p0 > 8 || P + 9 <= 48));
}

// Sort n unique strings.  Buffers: see SortBuffers.  Returns 0 or a hipError_t.
int dict_sort(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len, uint64_t n64,
              SortBuffers &b, hipStream_t s, uint32_t *rounds_out, uint32_t max_len, const uint64_t *uniq_key)
{
    const uint32_t n = (uint32_t)n64;
    uint32_t rounds = 0;
    if (n == 0) { if (rounds_out) *rounds_out = 0; return 0; }
    const uint32_t nb256 = (n + 255) / 256;
    hipLaunchKernelGGL(k_sort_init, dim3(nb256), dim3(256), 0, s, b.order, b.active_a, b.group, n);
    // scratch carved from flags: head[m], still[m], head_scan[m+1], still_scan[m+1], tmp[m], group_next[m]
    uint32_t *head = b.flags;
    uint32_t *still = head + n;
    uint32_t *head_scan = still + n;
    uint32_t *still_scan = head_scan + n + 1;
    uint32_t *tmp = still_scan + n + 1;
    uint32_t *group_next0 = tmp + n;
    // spare words behind the histograms (SORT_SPARE_WORDS): [0] runs longer than SMALL_RUN that stay tied after the round, [1] an
    // error flag, [2], [3] where the one long run lies in the next working set (k_sort_finish_small); [4..6] the classes' counts;
    // from [8]: the sample, then the packed key's table
    uint32_t *d_big = b.hist + 2 * ((size_t)256 * sort_blocks(n) + 1);
    uint64_t *d_sample = reinterpret_cast<uint64_t *>(d_big + 8);
    PackedBucket *d_table = reinterpret_cast<PackedBucket *>(d_sample + 4 * SORT_SAMPLE);

    // One list of positions [off0, off0 + m0) of `order` sorted: rounds of 8 bytes per string, tie groups, the finishing kernels.
    // identity: order[i] = i still holds (a sort of all ids from the start).  pk (with identity, over all ids): the first round is the
    // packed one -- it leaves holes | conforming strings | exceptions (the exceptions in no order yet), the rounds behind it go on
    // with the conforming strings still tied, at byte 8 * depth_after.
    std::function<int(uint32_t, uint32_t, bool, uint32_t)> sort_any;      // (below: a list sorted with a packed round where that pays)
    // (counts_out: the packed round's class counts, read back with the round's other figures -- before a long run can start a packed
    // round of its own, which uses the same words)
    auto sort_region = [&](uint32_t off0, uint32_t m0, bool identity, const PackedKeySpec *pk, uint32_t depth_after, uint32_t depth0,
                           uint32_t *counts_out) -> int {
        uint32_t *active = b.active_a, *active_alt = b.active_b;
        uint32_t *group = b.group, *group_next = group_next0;      // tie-group labels per active index
        uint32_t m = m0, off = off0, max_label = 0;   // max_label: the largest tie-group label of the round (0 in the first: one group)
        bool packed = pk != nullptr, first = true;
        HIPCHK(hipMemsetAsync(d_big, 0, 16, s));
        // the working set is [off, off + m) of the active list (off > 0: the region's first round, or one long run alone)
        for (uint32_t depth = depth0; m > 1; ++depth) {
            ++rounds;
            if (experiment("ZDW_SORT_DEBUG")) fprintf(stderr, "[sort] depth %u%s: m = %u of n = %u\n", depth, packed ? " (packed)" : "", m, n);
            const uint32_t gb = (m + 255) / 256;
            uint64_t *keys = b.key_a, *keys_alt = b.key_b;
            uint32_t *vals = b.val_a, *vals_alt = b.val_b;
            const uint32_t *order_in = (first && identity) ? nullptr : b.order;
            if (packed) hipLaunchKernelGGL(k_sort_keys_packed, dim3(std::min<uint32_t>(gb, 4096u)), dim3(256), 0, s, uniq_key, uniq_len, m, *pk, keys, vals, d_big + 4,
                                           order_in, order_in ? active + off : (const uint32_t *)nullptr);
            else hipLaunchKernelGGL(k_sort_keys, dim3(gb), dim3(256), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, order_in, active + off, m, depth, keys, vals, uniq_key);
            radix_sort_pairs(&keys, &vals, &keys_alt, &vals_alt, nullptr, m, 0, packed ? (pk->rest_bits + 10 + 7) & ~7u : 64u, b.hist, b.scan_tmp, s);
            if (!first && max_label > 1) {
                // stable, more significant key: the tie group of the element's old position (labels are < m + 1)
                // (labels are 1 .. the number of runs that stayed in the working set: read back with the round's count)
                uint32_t bits = 1;
                while (bits < 32 && (1ull << bits) <= (uint64_t)max_label) ++bits;
                radix_sort_pairs(&keys, &vals, &keys_alt, &vals_alt, group + off, m, 0, (bits + 7) & ~7u, b.hist, b.scan_tmp, s);
            }
            hipLaunchKernelGGL(k_sort_gather, dim3(gb), dim3(256), 0, s, order_in, active + off, vals, m, tmp, d_big);
            hipLaunchKernelGGL(k_sort_ties, dim3(gb), dim3(256), 0, s, b.order, active + off, group + off, tmp, (const uint64_t *)keys, m, still, d_big,
                               packed ? 1ull << (pk->rest_bits + 9) : 0ull);
            launch_scan_two_flags(still, still_scan, head_scan, m, b.scan_tmp, s);
            uint32_t *run_start = reinterpret_cast<uint32_t *>(keys == b.key_a ? b.key_b : b.key_a);      // (the pairs' other buffer: free behind the radix passes)
            hipLaunchKernelGGL(k_sort_compact, dim3(gb), dim3(256), 0, s, active + off, still, head_scan, still_scan, m, active_alt,
                               group_next, run_start);
            uint32_t m_next = 0, nbig = 0;
            HIPCHK(hipMemcpyAsync(&m_next, still_scan + m, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(&max_label, head_scan + m, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIPCHK(hipMemcpyAsync(&nbig, d_big, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            if (packed && counts_out) HIPCHK(hipMemcpyAsync(counts_out, d_big + 4, 12, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            uint32_t *t = active; active = active_alt; active_alt = t;
            t = group; group = group_next; group_next = t;
            m = m_next;
            off = 0;
            if (experiment("ZDW_SORT_DEBUG")) fprintf(stderr, "[sort]   -> %u still tied in %u runs, %u of them longer than %u\n", m, max_label, nbig, SMALL_RUN);
            if (m > 1 && nbig <= 1 && max_label > nbig && !test_hook("ZDW_SORT_NO_FINISH")) {
                // Every run that is still tied is small, or all but one are: a wave each finishes the small ones.  What remains is nothing,
                // or the one long run -- contiguous in the working set, so the next round takes it alone and needs no tie-group passes.
                // (The benchmark's table after its first round: 1.7 M date-times that all begin "2024-01-" among 0.3 M pairs and triples.)
                // (runs of a few strings: a thread per string; runs of dozens: a wave per run)
                if (nbig || m / max_label < 8)
                    hipLaunchKernelGGL(k_sort_finish_tiny, dim3((m + 255) / 256), dim3(256), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, uniq_key, b.order,
                                       active, group, run_start, max_label, m, tmp, d_big + 1);
                else
                    hipLaunchKernelGGL(k_sort_finish_small, dim3((max_label + 3) / 4), dim3(256), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, uniq_key, b.order,
                                       active, run_start, max_label, m, tmp, d_big + 1);
                hipLaunchKernelGGL(k_sort_place, dim3((m + 255) / 256), dim3(256), 0, s, b.order, active, tmp, m, nbig ? d_big + 2 : (const uint32_t *)nullptr);
                uint32_t res[3] = {0, 0, 0};
                HIPCHK(hipMemcpyAsync(res, d_big + 1, sizeof res, hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                if (res[0]) return (int)hipErrorUnknown;
                if (!nbig) break;
                if (res[2] <= res[1] || res[2] > m) return (int)hipErrorUnknown;
                off = res[1]; m = res[2] - res[1]; max_label = 1;
                // The one long run (the benchmark's table: 1.67 M date-times that all begin "2024-01-"): its strings lie side by side in
                // `order`, are tied over their first 8 (depth + 1) bytes and are of one kind -- a packed round of their own may cover
                // all that is left of them (sort_any).
                const uint32_t next = depth + (packed ? depth_after - depth : 1u);
                if (uniq_key && next < 3 && !test_hook("ZDW_SORT_NO_PACK") && (m >= (1u << 20) || test_hook("ZDW_SORT_PACK"))) {
                    uint32_t pos0 = 0;
                    HIPCHK(hipMemcpyAsync(&pos0, active + off, 4, hipMemcpyDeviceToHost, s));
                    HIPCHK(hipStreamSynchronize(s));
                    if ((uint64_t)pos0 + m > n) return (int)hipErrorUnknown;      // (cannot happen)
                    hipLaunchKernelGGL(k_sort_init_range, dim3((m + 255) / 256), dim3(256), 0, s, b.active_a, b.group, pos0, m);
                    return sort_any(pos0, m, false, next);
                }
            }
            if (depth > max_len / 8 + 2) return (int)hipErrorUnknown;   // cannot happen: two distinct strings differ within max_len bytes
            if (packed) { depth = depth_after - 1u; packed = false; }   // on behind the characters every first byte's packed key covered
            first = false;
        }
        return 0;
    };

    // A list of positions [pos0, pos0 + m0) of `order` (identity: all ids from the start, holes included; otherwise a run of strings
    // tied over their first 8 * depth0 bytes), sorted: with a packed round where the sample says it pays -- the packed round, the
    // rounds the conforming strings still need, the exceptions sorted apart, the two lists merged -- or by ordinary rounds.
    // (whole dictionaries: from 8 M strings on -- the sample and its verdict cost 0.1-0.3 ms; a long run: from 1 M)
    const bool force = test_hook("ZDW_SORT_PACK") != nullptr;
    sort_any = [&](uint32_t pos0, uint32_t m0, bool identity, uint32_t depth0) -> int {
        PackPlan pl;
        const uint32_t p0 = 8 * depth0;
        if (uniq_key && m0 > 1 && p0 < 24 && !test_hook("ZDW_SORT_NO_PACK") && (force || m0 >= (identity ? (8u << 20) : (1u << 20)))) {
            const uint32_t S = std::min(m0, identity ? SORT_SAMPLE : SORT_SAMPLE / 4);
            std::vector<uint64_t> sample(4 * (size_t)S);
            hipLaunchKernelGGL(k_sort_sample, dim3((S + 255) / 256), dim3(256), 0, s, uniq_key, uniq_len, m0, S, d_sample,
                               identity ? (const uint32_t *)nullptr : b.order + pos0);
            HIPCHK(hipMemcpyAsync(sample.data(), d_sample, sample.size() * 8, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            plan_packed_round(sample.data(), S, force, p0, pl);
            if (experiment("ZDW_SORT_DEBUG"))
                fprintf(stderr, "[sort] packed round at byte %u over %u strings: %s (up to byte %u in %u + 9 bits, %.2f %% of the sample left out, the next round at byte %u)\n",
                        p0, m0, pl.use ? "yes" : "no", pl.chars, pl.P, 100.0 * pl.exceptions, 8 * pl.depth_after);
        }
        if (!pl.use) return sort_region(pos0, m0, identity, nullptr, 0, depth0, nullptr);
        uint32_t *d_counts = d_big + 4;
        uint32_t counts[3] = {0, 0, 0};
        HIPCHK(hipMemcpyAsync(d_table, pl.table.data(), 256 * sizeof(PackedBucket), hipMemcpyHostToDevice, s));
        HIPCHK(hipMemsetAsync(d_counts, 0, 12, s));
        HIPCHK(hipStreamSynchronize(s));                            // (pl.table has been read)
        const PackedKeySpec ps{d_table, pl.P, p0};
        int e = sort_region(pos0, m0, identity, &ps, pl.depth_after, depth0, counts);
        if (e) return e;
        const uint32_t nh = counts[0], na = counts[1], nx = counts[2];
        if ((uint64_t)nh + na + nx != m0) return (int)hipErrorUnknown;      // (cannot happen)
        if (experiment("ZDW_SORT_DEBUG")) fprintf(stderr, "[sort] %u holes, %u conforming strings, %u exceptions\n", nh, na, nx);
        if (!nx) return 0;
        uint32_t *A = b.order + pos0 + nh, *B = A + na;
        if (nx > 1 && nx <= SORT_TINY) {
            hipLaunchKernelGGL(k_sort_tiny, dim3(1), dim3(256), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, uniq_key, B, nx, tmp);
            HIPCHK(hipMemcpyAsync(B, tmp, (size_t)nx * 4, hipMemcpyDeviceToDevice, s));
            ++rounds;
        } else if (nx > 1) {
            hipLaunchKernelGGL(k_sort_init_range, dim3((nx + 255) / 256), dim3(256), 0, s, b.active_a, b.group, pos0 + nh + na, nx);
            e = sort_region(pos0 + nh + na, nx, false, nullptr, 0, depth0, nullptr);
            if (e) return e;
        }
        if (na) {
            uint32_t *rank = still;
            hipLaunchKernelGGL(k_merge_ranks, dim3((nx + 255) / 256), dim3(256), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, uniq_key, A, na, B, nx, rank);
            hipLaunchKernelGGL(k_merge_place, dim3((na + nx + 255) / 256), dim3(256), 0, s, A, na, B, nx, rank, tmp);
            HIPCHK(hipMemcpyAsync(A, tmp, (size_t)(na + nx) * 4, hipMemcpyDeviceToDevice, s));
        }
        return 0;
    };
    const int e = sort_any(0, n, true, 0);
    if (e) return e;
    if (rounds_out) *rounds_out = rounds;
    return 0;
}

// ------------------------------------------------------------------------------------------
// offsets + blob
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dict_lengths(const uint32_t *__restrict__ order, const uint32_t *__restrict__ uniq_len,
                                                       uint32_t n, uint32_t *__restrict__ len_plus1)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { const uint32_t len = uniq_len[order ? order[r] : r]; len_plus1[r] = len ? len + 1 : 0u; }    // holes take no room; no order = identity
}

// Dictionary::getSize() - 1 in one launch: partial[b] = the sum of len + 1 over the strings block b looked at (holes take no room);
// the caller adds up the (at most DICT_TOTAL_BLOCKS) partial sums it reads back.
__global__ __launch_bounds__(256) void k_dict_total(const uint32_t *__restrict__ uniq_len, uint32_t n, uint64_t *__restrict__ partial)
{
    __shared__ uint64_t wsum[4];
    uint64_t v = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) { const uint32_t len = uniq_len[i]; v += len ? len + 1u : 0u; }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_down((int)(uint32_t)v, d, 64), hi = (uint32_t)__shfl_down((int)(uint32_t)(v >> 32), d, 64);
        v += ((uint64_t)hi << 32) | lo;
    }
    if ((threadIdx.x & 63u) == 0) wsum[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
uint32_t launch_dict_total(const uint32_t *uniq_len, uint64_t n, uint64_t *partial, hipStream_t s)
{
    uint32_t grid = (uint32_t)((n + 4095) / 4096);
    if (grid < 1) grid = 1;
    if (grid > DICT_TOTAL_BLOCKS) grid = DICT_TOTAL_BLOCKS;
    hipLaunchKernelGGL(k_dict_total, dim3(grid), dim3(256), 0, s, uniq_len, (uint32_t)n, partial);
    return grid;
}

void launch_dict_lengths(const uint32_t *order, const uint32_t *uniq_len, uint64_t n, uint32_t *len_plus1, hipStream_t s)
{
    if (!n) return;
    hipLaunchKernelGGL(k_dict_lengths, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, order, uniq_len, (uint32_t)n, len_plus1);
}

// one string per thread: copy into the blob (offset 1 + exclusive sum of len+1), NUL terminate,
// publish the offset for its unique id
__global__ __launch_bounds__(256) void k_dict_emit(const uint8_t *__restrict__ in, uint64_t in_bytes, const uint8_t *__restrict__ extra,
                                                    const uint32_t *__restrict__ order, const uint64_t *__restrict__ uniq_pos,
                                                    const uint32_t *__restrict__ uniq_len, const uint64_t *__restrict__ off_excl, uint32_t n,
                                                    uint8_t *__restrict__ blob, uint32_t *__restrict__ id_off,
                                                    const uint64_t *__restrict__ uniq_key)
{
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r == 0) blob[0] = 0;                                     // origin byte (dictionary.cpp:96)
    if (r >= n) return;
    const uint32_t u = order[r];
    // (its length from the offsets -- the scan's n + 1 outputs, read in sequence -- instead of another gather by id: 36 M of them per
    // block of C5)
    const uint64_t at = off_excl[r];
    const uint32_t len1 = (uint32_t)(off_excl[r + 1] - at);
    if (!len1) return;                                           // hole
    const uint32_t len = len1 - 1u;
    const uint64_t off = 1 + at;
    uint8_t *q = blob + off;
    id_off[u] = (uint32_t)off;
    if (uniq_key && len <= 24u) {
        // the whole string was kept by id when it entered the table (three words, zero beyond its end)
        const uint64_t *k = uniq_key + 3ull * u;
        uint64_t w = k[0];
        uint32_t i = 0;
        for (; i + 8 <= len; i += 8) {
            reinterpret_cast<PackedU64 *>(q + i)->v = w;
            w = i + 8 < len ? k[i / 8 + 1] : 0ull;
        }
        for (; i < len; ++i) { q[i] = (uint8_t)w; w >>= 8; }
        q[len] = 0;
        return;
    }
    const uint64_t pos = uniq_pos[u];
    const uint8_t *p = string_at(in, extra, pos);
    const uint8_t *end = pos >= EXTRA_BASE ? p + len + 16 : in + in_bytes;       // imported strings carry slack
    uint32_t i = 0;
    for (; i + 8 <= len; i += 8) reinterpret_cast<PackedU64 *>(q + i)->v = load_upto8(p + i, 8, end);     // unaligned 8-byte moves
    if (i < len) {
        uint64_t w = load_upto8(p + i, len - i, end);
        for (; i < len; ++i) { q[i] = (uint8_t)w; w >>= 8; }
    }
    q[len] = 0;
}

void launch_dict_emit(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint32_t *order, const uint64_t *uniq_pos,
                      const uint32_t *uniq_len, const uint64_t *off_excl, uint64_t n, uint8_t *blob, uint32_t *id_off, hipStream_t s,
                      const uint64_t *uniq_key)
{
    const uint32_t grid = n ? (uint32_t)((n + 255) / 256) : 1u;
    hipLaunchKernelGGL(k_dict_emit, dim3(grid), dim3(256), 0, s, in, in_bytes, extra, order, uniq_pos, uniq_len, off_excl, (uint32_t)n, blob, id_off, uniq_key);
}

// ------------------------------------------------------------------------------------------
// multi-GPU merge: strings travel as NUL-terminated bytes + u32 lengths
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_export_strings(const uint8_t *__restrict__ in, const uint8_t *__restrict__ extra,
                                                         const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len,
                                                         const uint32_t *__restrict__ order, const uint64_t *__restrict__ off_excl, uint32_t n,
                                                         uint8_t *__restrict__ bytes, uint32_t *__restrict__ lens_out,
                                                         const uint64_t *__restrict__ uniq_key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t u = order ? order[i] : i;
    const uint32_t len = uniq_len[u];
    if (lens_out) lens_out[i] = len;
    if (!len) return;                                            // hole: no bytes were reserved for it
    uint8_t *q = bytes + off_excl[i];
    if (uniq_key && len <= 24u) {                                // the whole string was kept by id (Pass1Args.uniq_key)
        const uint64_t *k = uniq_key + 3ull * u;
        uint64_t w = k[0];
        uint32_t j = 0;
        for (; j + 8 <= len; j += 8) {
            reinterpret_cast<PackedU64 *>(q + j)->v = w;
            w = j + 8 < len ? k[j / 8 + 1] : 0ull;
        }
        for (; j < len; ++j) { q[j] = (uint8_t)w; w >>= 8; }
        q[len] = 0;
        return;
    }
    const uint8_t *p = string_at(in, extra, uniq_pos[u]);
    for (uint32_t k = 0; k < len; ++k) q[k] = p[k];
    q[len] = 0;
}

void launch_export_strings(const uint8_t *in, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len, const uint32_t *order,
                           const uint64_t *off_excl, uint64_t n, uint8_t *bytes, uint32_t *lens_out, hipStream_t s, const uint64_t *uniq_key)
{
    if (!n) return;
    hipLaunchKernelGGL(k_export_strings, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, in, extra, uniq_pos, uniq_len, order, off_excl,
                       (uint32_t)n, bytes, lens_out, uniq_key);
}

// ------------------------------------------------------------------------------------------
// range-partitioned dictionary (multi-GPU): 16-byte prefix keys, splitter search, offset plumbing
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void key16_of(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos,
                                         const uint32_t *uniq_len, uint32_t u, uint64_t &k0, uint64_t &k1)
{
    const uint64_t pos = uniq_pos[u];
    const uint32_t len = uniq_len[u];
    const uint8_t *p = string_at(in, extra, pos);
    const uint8_t *end = pos >= EXTRA_BASE ? p + len + 16 : in + in_bytes;
    k0 = sort_key8(p, len, 0, end);
    k1 = sort_key8(p, len, 1, end);
}

// keys[2i], keys[2i+1] = first 16 bytes (big-endian words) of the string at sorted position idx[i]
__global__ __launch_bounds__(64) void k_key16_at(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos,
                                                  const uint32_t *uniq_len, const uint32_t *order, const uint32_t *idx, uint32_t n, uint64_t *keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key16_of(in, in_bytes, extra, uniq_pos, uniq_len, order[idx[i]], keys[2 * i], keys[2 * i + 1]);
}

// the same for unsorted uniques: sample i = the first real string at or after id idx[i] (within 256 ids; ~0 when none)
__global__ __launch_bounds__(64) void k_key16_sample(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos,
                                                      const uint32_t *uniq_len, const uint32_t *idx, uint32_t n, uint32_t nids, uint64_t *keys)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t id = idx[i];
    for (uint32_t k = 0; k < 256 && id < nids && uniq_len[id] == 0; ++k) ++id;
    if (id >= nids || uniq_len[id] == 0) { keys[2 * i] = keys[2 * i + 1] = ~0ull; return; }
    key16_of(in, in_bytes, extra, uniq_pos, uniq_len, id, keys[2 * i], keys[2 * i + 1]);
}

// bucket of every unique id for the range partition: 0 = hole, 1 + (number of splitters <= the string's 16-byte key)
__global__ __launch_bounds__(256) void k_bucket_of(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *__restrict__ uniq_pos,
                                                    const uint32_t *__restrict__ uniq_len, uint32_t nids, const uint64_t *__restrict__ splitters,
                                                    uint32_t nsplit, uint64_t *__restrict__ keys, uint32_t *__restrict__ vals,
                                                    const uint64_t *__restrict__ uniq_key)
{
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nids) return;
    uint32_t b = 0;
    if (uniq_len[id]) {
        uint64_t k0, k1;
        if (uniq_key) { k0 = __builtin_bswap64(uniq_key[3ull * id]); k1 = __builtin_bswap64(uniq_key[3ull * id + 1]); }      // (kept by id: a streaming read)
        else key16_of(in, in_bytes, extra, uniq_pos, uniq_len, id, k0, k1);
        uint32_t lo = 0, hi = nsplit;          // first splitter greater than the key
        while (lo < hi) {
            const uint32_t m = (lo + hi) >> 1;
            const uint64_t s0 = splitters[2 * m], s1 = splitters[2 * m + 1];
            const bool greater = s0 > k0 || (s0 == k0 && s1 > k1);
            if (greater) hi = m; else lo = m + 1;
        }
        b = 1 + lo;
    }
    keys[id] = b;
    vals[id] = id;
}

// bounds[j] = first position whose key is >= j, for j = 0 .. nq - 1 (keys ascending)
__global__ __launch_bounds__(64) void k_lower_bounds(const uint64_t *__restrict__ keys, uint32_t n, uint32_t nq, uint32_t *__restrict__ bounds)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nq) return;
    uint32_t a = 0, b = n;
    while (a < b) {
        const uint32_t m = a + (b - a) / 2;
        if (keys[m] < j) a = m + 1; else b = m;
    }
    bounds[j] = a;
}

// bounds[j] = first sorted position p in [lo, hi) whose 16-byte key is >= splitter j (hi when none)
__global__ __launch_bounds__(64) void k_partition_search(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos,
                                                          const uint32_t *uniq_len, const uint32_t *order, uint32_t lo, uint32_t hi,
                                                          const uint64_t *splitters, uint32_t nsplit, uint32_t *bounds)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nsplit) return;
    const uint64_t s0 = splitters[2 * j], s1 = splitters[2 * j + 1];
    uint32_t a = lo, b = hi;
    while (a < b) {
        const uint32_t m = a + (b - a) / 2;
        uint64_t k0, k1;
        key16_of(in, in_bytes, extra, uniq_pos, uniq_len, order[m], k0, k1);
        const bool less = k0 < s0 || (k0 == s0 && k1 < s1);
        if (less) a = m + 1; else b = m;
    }
    bounds[j] = a;
}

__global__ __launch_bounds__(256) void k_owner_offsets(const uint32_t *__restrict__ ids, const uint32_t *__restrict__ id_off, uint32_t base,
                                                        uint32_t n, uint32_t *__restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    // (a string the owner dropped -- a hole of the layout, id_off 0: local offsets start at 1 -- has no offset: 0, never base + garbage)
    if (i < n) {
        const uint32_t o = ids[i] ? id_off[ids[i] - 1] : 0u;
        out[i] = o ? base + o : 0u;
    }
}

__global__ __launch_bounds__(256) void k_scatter_offsets(const uint32_t *__restrict__ order, const uint32_t *__restrict__ offsets, uint32_t n,
                                                          uint32_t *__restrict__ id_off)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) id_off[order[i]] = offsets[i];
}

void launch_key16_at(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                     const uint32_t *order, const uint32_t *idx, uint32_t n, uint64_t *keys, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_key16_at, dim3((n + 63) / 64), dim3(64), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, order, idx, n, keys);
}
void launch_key16_sample(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                         const uint32_t *idx, uint32_t n, uint32_t nids, uint64_t *keys, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_key16_sample, dim3((n + 63) / 64), dim3(64), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, idx, n, nids, keys);
}
// Groups the unique ids by range: order[0 .. bounds[1]) = holes, order[bounds[p + 1] .. bounds[p + 2]) = the strings of part p
// (ids ascending inside a group); bounds has nparts + 2 entries (device).  One stable radix pass on the bucket number.
void launch_bucket_partition(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                             uint32_t nids, const uint64_t *splitters, uint32_t nparts, SortBuffers &b, uint32_t *order_out, uint32_t *bounds,
                             hipStream_t s, const uint64_t *uniq_key)
{
    if (!nids) { (void)hipMemsetAsync(bounds, 0, (nparts + 2) * 4, s); return; }
    uint64_t *keys = b.key_a, *keys_alt = b.key_b;
    uint32_t *vals = b.val_a, *vals_alt = b.val_b;
    hipLaunchKernelGGL(k_bucket_of, dim3((nids + 255) / 256), dim3(256), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, nids, splitters, nparts - 1,
                       keys, vals, uniq_key);
    radix_sort_pairs(&keys, &vals, &keys_alt, &vals_alt, nullptr, nids, 0, nparts + 1 <= 256 ? 8 : 16, b.hist, b.scan_tmp, s);
    hipLaunchKernelGGL(k_lower_bounds, dim3((nparts + 2 + 63) / 64), dim3(64), 0, s, (const uint64_t *)keys, nids, nparts + 2, bounds);
    (void)hipMemcpyAsync(order_out, vals, (size_t)nids * 4, hipMemcpyDeviceToDevice, s);
}
void launch_partition_search(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                             const uint32_t *order, uint32_t lo, uint32_t hi, const uint64_t *splitters, uint32_t nsplit, uint32_t *bounds,
                             hipStream_t s)
{
    if (nsplit) hipLaunchKernelGGL(k_partition_search, dim3((nsplit + 63) / 64), dim3(64), 0, s, in, in_bytes, extra, uniq_pos, uniq_len, order,
                                   lo, hi, splitters, nsplit, bounds);
}
void launch_owner_offsets(const uint32_t *ids, const uint32_t *id_off, uint32_t base, uint64_t n, uint32_t *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_owner_offsets, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, ids, id_off, base, (uint32_t)n, out);
}
void launch_scatter_offsets(const uint32_t *order, const uint32_t *offsets, uint64_t n, uint32_t *id_off, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_scatter_offsets, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, order, offsets, (uint32_t)n, id_off);
}

// ------------------------------------------------------------------------------------------
// The reference's block limit as a deterministic model (DESIGN.md states the fit): a block ends
// when StringHeap takes a new 64 MiB arena while the process is at or beyond --mem-limit.  The strings enter the
// arenas in the order pass 1 first meets them, so: unique ids by position, prefix sums of len + 1, then a walk over
// the arenas (a few dozen binary searches) finds the string whose arena trips the limit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_model_keys(const uint64_t *__restrict__ uniq_pos, const uint32_t *__restrict__ uniq_len, uint32_t n,
                                                     uint64_t *__restrict__ keys, uint32_t *__restrict__ vals)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = uniq_len[i] ? (uniq_pos[i] & POS_MASK) : POS_MASK;      // holes (length 0) after every string
    vals[i] = i;
}

// one thread: arenas[0] = index (in position order) of the string whose arena allocation trips the limit, or ~0
__global__ void k_model_trip(const uint64_t *__restrict__ excl /* n + 1 */, uint32_t nreal, double base_mb, double node_bytes, double limit_mb,
                             uint64_t arena_bytes, uint64_t *__restrict__ out)
{
    uint64_t start = 0, arenas = 0, trip = ~0ull;
    while (start < nreal) {
        ++arenas;
        const double vsize = base_mb + 64.0 * (double)arenas + node_bytes * (double)(start + 1) / 1048576.0;
        if (vsize >= limit_mb) { trip = start; break; }
        const uint64_t need0 = excl[start + 1] - excl[start];
        const uint64_t cap = need0 > arena_bytes ? need0 : arena_bytes;
        // the arena holds strings start .. j - 1: the largest j with excl[j] - excl[start] <= cap
        uint64_t lo = start + 1, hi = nreal;              // invariant: excl[lo] - excl[start] <= cap
        const uint64_t lim = excl[start] + cap;
        while (lo < hi) {
            const uint64_t m = lo + (hi - lo + 1) / 2;
            if (excl[m] <= lim) lo = m; else hi = m - 1;
        }
        start = lo;
    }
    out[0] = trip;
    out[1] = arenas;
}

// Where in its row does position p lie?  One wave: out[0] = start of the row that holds p, out[1] = unescaped tabs
// between the row's start and p (= the field's column).
__global__ __launch_bounds__(64) void k_locate_field(const uint8_t *__restrict__ in, uint64_t lo, uint64_t p, uint64_t *__restrict__ out)
{
    const uint32_t lane = threadIdx.x;
    uint64_t rs = lo;
    for (uint64_t hi = p; hi > lo;) {                     // backwards, 64 bytes at a time: the last row terminator before p
        const uint64_t q = hi - 1 - lane;
        const bool t = hi > lo + lane && in[q] == '\n' && is_row_terminator(in, lo, q);
        const uint64_t m = __ballot(t);
        if (m) { rs = hi - (uint64_t)lowest_bit(m); break; }
        hi = hi > lo + 64 ? hi - 64 : lo;
    }
    uint32_t tabs = 0;
    for (uint64_t q = rs + lane; q < p; q += 64)
        if (in[q] == '\t' && (backslashes_before(in, rs, q) & 1u) == 0) ++tabs;
    uint32_t total;
    wave_excl_sum(tabs, lane, total);
    if (lane == 0) { out[0] = rs; out[1] = total; }
}

// ---- the same walk over the shards of a row-sharded block (zdw_shard_model_walk) ----
// The block's strings enter the reference's arenas in the order of their first occurrence: shard 0's, then those of shard 1 that
// shard 0 does not hold, ...  Every shard walks its own node-wide-first strings from the state the shard before it left:
// state[0] arenas taken, state[1] dictionary entries, state[2] bytes left in the open arena.  out[0] = index (in position order) of
// the string whose arena trips the limit, or ~0; out[1..3] = the state behind this shard's strings.
__global__ void k_model_trip_chained(const uint64_t *__restrict__ excl /* n + 1 */, uint32_t n, uint64_t arenas0, uint64_t entries0, uint64_t free0,
                                     double base_mb, double node_bytes, double limit_mb, uint64_t arena_bytes, uint64_t *__restrict__ out)
{
    uint64_t start = 0, arenas = arenas0, trip = ~0ull, left = free0;
    if (n && free0) {                                    // what still fits the arena that is open: the largest j with excl[j] <= free0
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t m = lo + (hi - lo + 1) / 2;
            if (excl[m] - excl[0] <= free0) lo = m; else hi = m - 1;
        }
        start = lo;
        left = free0 - (excl[lo] - excl[0]);
    }
    while (start < n) {
        ++arenas;
        const double vsize = base_mb + 64.0 * (double)arenas + node_bytes * (double)(entries0 + start + 1) / 1048576.0;
        if (vsize >= limit_mb) { trip = start; break; }
        const uint64_t need0 = excl[start + 1] - excl[start];
        const uint64_t cap = need0 > arena_bytes ? need0 : arena_bytes;
        uint64_t lo = start + 1, hi = n;
        const uint64_t lim = excl[start] + cap;
        while (lo < hi) {
            const uint64_t m = lo + (hi - lo + 1) / 2;
            if (excl[m] <= lim) lo = m; else hi = m - 1;
        }
        left = lim - excl[lo];
        start = lo;
    }
    out[0] = trip;
    out[1] = arenas;
    out[2] = entries0 + n;
    out[3] = left;
}
void launch_model_trip_chained(const uint64_t *excl, uint32_t n, uint64_t arenas0, uint64_t entries0, uint64_t free0, double base_mb, double node_bytes,
                               double limit_mb, uint64_t arena_bytes, uint64_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_model_trip_chained, dim3(1), dim3(1), 0, s, excl, n, arenas0, entries0, free0, base_mb, node_bytes, limit_mb, arena_bytes, out);
}

// lengths of this shard's strings as the walk sees them: a string some earlier shard holds too is none of this shard's (length 0).
// order[j] = unique id of the j-th exported string, flags[j] = the owner's verdict on it; mlen is zeroed beforehand; *count += firsts
__global__ __launch_bounds__(256) void k_mask_first(const uint32_t *__restrict__ order, const uint32_t *__restrict__ flags, uint32_t n,
                                                     const uint32_t *__restrict__ uniq_len, uint32_t *__restrict__ mlen, uint32_t *__restrict__ count)
{
    // (a workgroup walks its share of the strings and adds its count once: one atomic per wave on one word is what k_trim_ids lost
    // 6.7 ms to)
    __shared__ uint32_t firsts_of_wave[4];
    uint32_t firsts = 0;
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (uint64_t)gridDim.x * 256) {
        if (!flags[j]) continue;
        const uint32_t id = order[j];
        const uint32_t l = uniq_len[id];
        if (l) { mlen[id] = l; ++firsts; }
    }
    firsts = wave_incl_sum(firsts);
    if (lane_id() == 63) firsts_of_wave[threadIdx.x >> 6] = firsts;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t k = firsts_of_wave[0] + firsts_of_wave[1] + firsts_of_wave[2] + firsts_of_wave[3];
        if (k) atomicAdd(count, k);
    }
}
void launch_mask_first(const uint32_t *order, const uint32_t *flags, uint32_t n, const uint32_t *uniq_len, uint32_t *mlen, uint32_t *count, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_mask_first, dim3(std::min<uint32_t>((n + 255) / 256, 2048u)), dim3(256), 0, s, order, flags, n, uniq_len, mlen, count);
}

// position of the first occurrence of every exported string (order[j] = its unique id), `base` added
__global__ __launch_bounds__(256) void k_export_positions(const uint32_t *__restrict__ order, const uint64_t *__restrict__ uniq_pos, uint32_t n, uint64_t base,
                                                           uint64_t *__restrict__ out)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = base + (uniq_pos[order[j]] & POS_MASK);
}
void launch_export_positions(const uint32_t *order, const uint64_t *uniq_pos, uint32_t n, uint64_t base, uint64_t *out, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_export_positions, dim3((n + 255) / 256), dim3(256), 0, s, order, uniq_pos, n, base, out);
}

// owner side: the smallest position any shard reported for a string (ids[j] = unique id + 1 of arrival j, 0 = none)
__global__ __launch_bounds__(256) void k_owner_minpos(const uint32_t *__restrict__ ids, const uint64_t *__restrict__ pos, uint32_t n, uint64_t *__restrict__ minpos)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n && ids[j]) atomicMin(reinterpret_cast<unsigned long long *>(minpos + (ids[j] - 1u)), (unsigned long long)pos[j]);
}
__global__ __launch_bounds__(256) void k_owner_first(const uint32_t *__restrict__ ids, const uint64_t *__restrict__ pos, uint32_t n, const uint64_t *__restrict__ minpos,
                                                      uint32_t *__restrict__ flags)
{
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) flags[j] = (ids[j] && pos[j] == minpos[ids[j] - 1u]) ? 1u : 0u;
}
void launch_owner_minpos(const uint32_t *ids, const uint64_t *pos, uint64_t n, uint64_t *minpos, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_owner_minpos, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, ids, pos, (uint32_t)n, minpos);
}
void launch_owner_first(const uint32_t *ids, const uint64_t *pos, uint64_t n, const uint64_t *minpos, uint32_t *flags, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_owner_first, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, ids, pos, (uint32_t)n, minpos, flags);
}
// strings first met beyond last_pos are none of this block's: holes
__global__ __launch_bounds__(256) void k_owner_filter(const uint64_t *__restrict__ minpos, uint32_t *__restrict__ ulen, uint32_t n, uint64_t last_pos)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && ulen[i] && minpos[i] > last_pos) ulen[i] = 0;
}
void launch_owner_filter(const uint64_t *minpos, uint32_t *ulen, uint32_t n, uint64_t last_pos, hipStream_t s)
{
    if (n) hipLaunchKernelGGL(k_owner_filter, dim3((n + 255) / 256), dim3(256), 0, s, minpos, ulen, n, last_pos);
}

void launch_model_sort(const uint64_t *uniq_pos, const uint32_t *uniq_len, uint32_t n, SortBuffers &b, uint32_t **vals_out, hipStream_t s)
{
    uint64_t *keys = b.key_a, *keys_alt = b.key_b;
    uint32_t *vals = b.val_a, *vals_alt = b.val_b;
    hipLaunchKernelGGL(k_model_keys, dim3((n + 255) / 256), dim3(256), 0, s, uniq_pos, uniq_len, n, keys, vals);
    radix_sort_pairs(&keys, &vals, &keys_alt, &vals_alt, nullptr, n, 0, (uint32_t)POS_BITS, b.hist, b.scan_tmp, s);
    *vals_out = vals;
}
void launch_model_trip(const uint64_t *excl, uint32_t nreal, double base_mb, double node_bytes, double limit_mb, uint64_t arena_bytes, uint64_t *out,
                       hipStream_t s)
{
    hipLaunchKernelGGL(k_model_trip, dim3(1), dim3(1), 0, s, excl, nreal, base_mb, node_bytes, limit_mb, arena_bytes, out);
}
void launch_locate_field(const uint8_t *in, uint64_t lo, uint64_t p, uint64_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(k_locate_field, dim3(1), dim3(64), 0, s, in, lo, p, out);
}

// ---- a block cut back to the row and field where the model ends it (zdw_capi.hip: trim_block) ----
// Pass 1 ran over more rows than the block keeps.  What the block's rows -- [0, rows) and the first `partial` fields of row `rows`
// -- leave behind is rebuilt from their records: the strings they refer to (every other unique id becomes a hole), the columns they
// fill, the ranges of their numbers.  A record holds what pass 2 writes; the range takes a char(1) field's first byte alone unless
// that byte is a backslash (zdw_field_dev.h, ConvertToZDW.cpp:358-361 against :543-547).
// MARK: the ids the kept records hold are marked in `live` (otherwise k_trim_ids goes by the strings' first positions and this kernel
// only reads: eight rows' loads in flight per thread).
template <bool MARK>
__global__ __launch_bounds__(256) void k_trim_scan(const uint8_t *__restrict__ rec, uint32_t rec_stride, uint32_t ncols, const uint8_t *__restrict__ col_kind,
                                                   const uint32_t *__restrict__ col_recoff, uint32_t rows, uint32_t partial, uint8_t *__restrict__ live,
                                                   uint64_t *__restrict__ col_min, uint64_t *__restrict__ col_max, uint8_t *__restrict__ col_seen,
                                                   const uint32_t *__restrict__ nz, uint32_t nz_stride)
{
    constexpr uint32_t TILE = 32;
    const uint64_t r0 = (uint64_t)blockIdx.x * TILE;
    const uint64_t last = (uint64_t)rows + (partial ? 1u : 0u);
    const uint64_t r1 = r0 + TILE < last ? r0 + TILE : last;
    for (uint32_t col = threadIdx.x; col < ncols; col += 256) {
        const uint32_t kind = col_kind[col], off = col_recoff[col];
        const uint64_t stop = (r1 > rows && col >= partial) ? rows : r1;       // the partial row: its first fields only
        uint64_t mn = ~0ull, mx = 0;
        bool seen = false;
        if (!MARK && !nz) {
            const uint8_t *p = rec + r0 * rec_stride + off;
            if (kind == K_TEXT) {
                uint32_t any = 0;
#pragma unroll 8
                for (uint64_t r = r0; r < stop; ++r, p += rec_stride) any |= *reinterpret_cast<const uint32_t *>(p);
                seen = any != 0;
            } else {
#pragma unroll 8
                for (uint64_t r = r0; r < stop; ++r, p += rec_stride) {
                    uint64_t v = *reinterpret_cast<const uint64_t *>(p);
                    if (kind == K_CHAR && (v & 0xffu) != (uint64_t)'\\') v = (uint64_t)(int64_t)(int8_t)(v & 0xffu);
                    mn = v && v < mn ? v : mn; mx = v > mx ? v : mx;
                }
            }
        } else {
            for (uint64_t r = r0; r < stop; ++r) {
                if (nz && !((nz[r * nz_stride + (col >> 5)] >> (col & 31u)) & 1u)) continue;           // (a matrix written through the map of non-zero records: zero elsewhere)
                const uint8_t *p = rec + r * rec_stride + off;
                if (kind == K_TEXT) {
                    const uint32_t id1 = *reinterpret_cast<const uint32_t *>(p);
                    if (id1) { if (MARK) live[id1 - 1u] = 1; seen = true; }
                } else {
                    uint64_t v = *reinterpret_cast<const uint64_t *>(p);
                    if (kind == K_CHAR && (v & 0xffu) != (uint64_t)'\\') v = (uint64_t)(int64_t)(int8_t)(v & 0xffu);
                    if (v) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
                }
            }
        }
        if (kind == K_TEXT) {
            if (seen && !col_seen[col]) col_seen[col] = 1;
        } else if (mx) {
            if (mn < col_min[col]) atomicMin(reinterpret_cast<unsigned long long *>(&col_min[col]), (unsigned long long)mn);
            if (mx > col_max[col]) atomicMax(reinterpret_cast<unsigned long long *>(&col_max[col]), (unsigned long long)mx);
        }
    }
}
// unique ids no kept row refers to become holes; *entries = the strings that stay.  Which ones stay: `live` (k_trim_scan marked the
// ids the kept records hold), or -- live == nullptr, the block was parsed with Pass1Args.track_first -- those whose first occurrence
// lies at or before last_pos, the tripping field's start (a kept field starts at or before it, any other behind it).
// A workgroup adds what it kept with one atomic (one per wave: 600 000 of them on one word took 6.7 of C5's 72 ms per block).
__global__ __launch_bounds__(256) void k_trim_ids(const uint8_t *__restrict__ live, const uint64_t *__restrict__ uniq_pos, uint64_t last_pos,
                                                  uint32_t *__restrict__ uniq_len, uint32_t n, uint32_t *__restrict__ entries)
{
    __shared__ uint32_t kept_of_wave[4];
    uint32_t kept = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) {
        if (!uniq_len[i]) continue;
        const bool keep = live ? live[i] != 0 : (uniq_pos[i] & POS_MASK) <= last_pos;
        if (keep) ++kept; else uniq_len[i] = 0;
    }
    kept = wave_incl_sum(kept);
    if (lane_id() == 63) kept_of_wave[threadIdx.x >> 6] = kept;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t k = kept_of_wave[0] + kept_of_wave[1] + kept_of_wave[2] + kept_of_wave[3];
        if (k) atomicAdd(entries, k);
    }
}
void launch_trim_block(const uint8_t *rec, uint32_t rec_stride, uint32_t ncols, const uint8_t *col_kind, const uint32_t *col_recoff, uint32_t rows,
                       uint32_t partial, uint8_t *live, const uint64_t *uniq_pos, uint64_t last_pos, uint64_t *col_min, uint64_t *col_max,
                       uint8_t *col_seen, uint32_t *uniq_len, uint32_t nids, uint32_t *entries, const uint32_t *nz, uint32_t nz_stride, hipStream_t s)
{
    const uint64_t last = (uint64_t)rows + (partial ? 1u : 0u);
    if (last && live) hipLaunchKernelGGL(k_trim_scan<true>, dim3((uint32_t)((last + 31) / 32)), dim3(256), 0, s, rec, rec_stride, ncols, col_kind, col_recoff, rows,
                                         partial, live, col_min, col_max, col_seen, nz, nz_stride);
    else if (last) hipLaunchKernelGGL(k_trim_scan<false>, dim3((uint32_t)((last + 31) / 32)), dim3(256), 0, s, rec, rec_stride, ncols, col_kind, col_recoff, rows,
                                      partial, live, col_min, col_max, col_seen, nz, nz_stride);
    if (nids) hipLaunchKernelGGL(k_trim_ids, dim3(std::min<uint32_t>((nids + 255) / 256, 2048u)), dim3(256), 0, s, live, uniq_pos, last_pos, uniq_len, nids, entries);
}

// Persistent waves: a wave keeps its chunk of unique ids across the groups of 64 strings it handles (one
// returning atomic on the id counter per wave and group would bound the kernel: ~30 M/s on one word).
__global__ __launch_bounds__(256) void k_import_strings(const uint8_t *in, uint64_t n, const uint8_t *extra, const uint32_t *__restrict__ lens,
                                                         const uint64_t *__restrict__ offs, uint64_t count, DictTable d, uint32_t max_entries,
                                                         uint32_t trim, uint64_t hi, uint32_t *__restrict__ ids_out)
{
    const uint32_t lane = lane_id();
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    IdAlloc al;
    al.cur = al.end = al.taken = 0;
    uint32_t created = 0;
    for (uint64_t g = wave * 64; g < count; g += nwaves * 64) {
        const uint64_t i = g + lane;
        bool is_new = false;
        uint32_t id1 = 0;
        const bool mine = i < count && lens[i] != 0;                 // a length of 0 is the exporter's hole
        const uint64_t mm = __ballot(mine);
        if (mm) {
            if (mine) {
                const uint8_t *p = extra + offs[i];
                id1 = dict_insert(p, p + lens[i] + 16, in, n, extra, hi, d, al, ID_CHUNK, EXTRA_BASE + offs[i], lens[i], trim != 0, is_new);
            }
            adopt_ids(al, mm);
        }
        if (ids_out && i < count) ids_out[i] = id1;                  // unique id + 1 of every imported string (0: hole)
        created += popc64(__ballot(is_new));
    }
    retire_ids(al, d.uniq_len, d.id_cap);
    if (created && lane == 0) {
        const uint32_t total = atomicAdd(&d.st->entries, created) + created;
        if (total > max_entries) d.st->overflow = 1;
    }
}

void launch_import_strings(const uint8_t *in, uint64_t n, const uint8_t *extra, const uint32_t *lens, const uint64_t *offs, uint64_t count,
                           uint64_t *slots, uint64_t slot_mask, uint32_t max_entries, uint64_t *uniq_pos, uint32_t *uniq_len, uint32_t id_cap,
                           uint32_t trim, uint64_t hi, P1Status *st, uint32_t *ids_out, hipStream_t s, uint64_t *uniq_key)
{
    if (!count) return;
    DictTable d;
    d.slots = slots; d.slot_mask = slot_mask; d.uniq_pos = uniq_pos; d.uniq_len = uniq_len; d.id_cap = id_cap; d.uniq_key = uniq_key; d.st = st; d.track_first = 0;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t want = (count + 255) / 256;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)cus * 8);
    hipLaunchKernelGGL(k_import_strings, dim3(grid), dim3(256), 0, s, in, n, extra, lens, offs, count, d, max_entries, trim, hi, ids_out);
}

}  // namespace zdw
