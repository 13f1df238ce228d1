// zdw_decode.hip — one ZDW block back to its TSV rows on gfx950: kernels and the zdw_dec_* C ABI.
//
// What the reference does row by row in UnconvertFromZDW<T>::readNextRow (cplusplus/UnconvertFromZDW.cpp:1270-1462): flag
// bytes, then the values of the flagged columns; a column whose flag is clear repeats the row above (0 before a block's first
// row); text-like values are offsets into the block's dictionary, integers are value + base, char(1) holds its byte(s).
//
// Where a row starts is a chain in this format: a row's length is the sum of the sizes of its flagged columns, known once
// its flag bytes are read, and there is no row index.  The chain is made parallel by speculation that is then proved:
//   walk<0>   every chunk of the row section is walked by one wave from half a MiB BEFORE its start: chains of different phase
//             fall into step after a few hundred rows (each treats data as flags and takes pseudo-random steps); the wave
//             leaves where its chain first reaches the next chunk -- a guess;
//   walk<1>   every chunk is walked again from the exit of the chunk before (chunk 0: from 0, the truth) and leaves its
//             exit and its number of rows.  If every exit equals the guess, induction from chunk 0 makes them all true; a chunk
//             whose guess was wrong (data on which chains never meet, e.g. a table whose rows all repeat the row above: rows of
//             nothing but zero flag bytes) gets its entry corrected and the pass is repeated -- at worst once per chunk;
//   walk<2>   writes the row starts at the rows' global indices (prefix sum of the counts).
// With the starts known the rest is parallel: expand (wave = row: values of flagged columns into a dense matrix), the value
// "before" each tile of 64 rows per column (last flagged value per tile, carried across tiles in three small passes), text
// lengths per row, prefix sum, text.
#include "../../include/zdw_hip.h"
#include "zdw_knobs.h"
#include "zdw_dev.h"
#include "zdw_kernels.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

namespace zdw {

extern __shared__ __attribute__((aligned(16))) uint8_t zdw_lds[];

namespace {

__constant__ uint64_t DEC_POW10[20] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull, 1000000000ull,
                                       10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull, 100000000000000ull,
                                       1000000000000000ull, 10000000000000000ull, 100000000000000000ull, 1000000000000000000ull,
                                       10000000000000000000ull};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v, uint32_t lane)
{
    uint32_t total;
    (void)wave_excl_sum(v, lane, total);
    return total;
}

// A wave's view of the row section: two consecutive 8 KiB blocks in an LDS ring, the block after them on its way in registers
// (requested when the walk enters the second block, put into the ring when it enters the third): the chain's steps read their
// flag bytes from LDS and the memory latency of a block is paid while the block before it is walked.
constexpr uint32_t DEC_BLK = 8192, DEC_BLK_SHIFT = 13;

struct DecWindow {
    const uint8_t *rows;
    uint64_t limit;          // bytes of the section that exist (16-byte reads up to 64 bytes past it stay inside the allocation)
    uint32_t lds;            // byte offset of the ring (2 x DEC_BLK) in zdw_lds
    uint64_t cur;            // blocks cur and cur + 1 are in the ring
    uint4 next[8];           // block cur + 2
};
__device__ __forceinline__ void dec_win_request(DecWindow &w, uint64_t blk, uint32_t lane)
{
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) {
        const uint64_t q = (blk << DEC_BLK_SHIFT) + (uint64_t)(i * 64u + lane) * 16u;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < w.limit) __builtin_memcpy(&v, w.rows + q, 16);                       // (the section starts at any byte: an alignment-1 load, which gfx950 takes as one dwordx4)
        w.next[i] = v;
    }
}
__device__ __forceinline__ void dec_win_commit(DecWindow &w, uint64_t blk, uint32_t lane)
{
    uint8_t *slot = zdw_lds + w.lds + (uint32_t)(blk & 1ull) * DEC_BLK;
#pragma unroll
    for (uint32_t i = 0; i < 8; ++i) *reinterpret_cast<uint4 *>(slot + (i * 64u + lane) * 16u) = w.next[i];
}
__device__ __forceinline__ void dec_win_seek(DecWindow &w, uint64_t p, uint32_t lane)
{
    const uint64_t b = p >> DEC_BLK_SHIFT;
    if (b == w.cur) return;
    if (b == w.cur + 1) {                      // the common step: the block in flight has long arrived
        dec_win_commit(w, w.cur + 2, lane);
        w.cur = b;
    } else {                                   // far away (the first call, a row longer than a block)
        dec_win_request(w, b, lane);
        dec_win_commit(w, b, lane);
        dec_win_request(w, b + 1, lane);
        dec_win_commit(w, b + 1, lane);
        w.cur = b;
    }
    dec_win_request(w, w.cur + 2, lane);
    __builtin_amdgcn_wave_barrier();
}

// bytes of values the flag bytes at p announce, + the flag bytes themselves: the row's length
template <bool TBL_LDS>
__device__ __forceinline__ uint32_t dec_row_len(DecWindow &w, uint64_t p, uint32_t nflag, const uint16_t *__restrict__ tbl, uint32_t lane)
{
    uint32_t sum = 0;
    if (TBL_LDS) {
        dec_win_seek(w, p, lane);
        for (uint32_t j = lane; j < nflag; j += 64) {
            const uint32_t b = zdw_lds[w.lds + (uint32_t)((p + j) & (2u * DEC_BLK - 1u))];
            sum += reinterpret_cast<const uint16_t *>(zdw_lds)[j * 256u + b];
        }
    } else {                                   // schemas of more than 512 used columns: flag bytes and table from memory
        for (uint32_t j = lane; j < nflag; j += 64) {
            const uint64_t q = p + j;
            const uint32_t b = q < w.limit ? w.rows[q] : 0u;
            sum += tbl[j * 256u + b];
        }
    }
    return nflag + wave_sum(sum, lane);
}

// MODE 0: guess (from `warm` bytes before the chunk); 1: from the entry: exit and rows; 2: from the entry: the rows' starts
template <int MODE, bool TBL_LDS>
__global__ __launch_bounds__(64) void k_dec_walk(const uint8_t *__restrict__ rows, uint64_t limit, uint32_t nflag, const uint16_t *__restrict__ tbl,
                                                 uint32_t chunk_shift, uint32_t nchunks, uint64_t warm, const uint64_t *__restrict__ entry_of,
                                                 uint64_t *__restrict__ exit_out, uint32_t *__restrict__ count_out,
                                                 const uint64_t *__restrict__ rowbase, uint64_t *__restrict__ rstart, uint64_t max_rows)
{
    const uint32_t lane = threadIdx.x;
    DecWindow w;
    w.rows = rows; w.limit = limit; w.cur = ~0ull - 8;
    w.lds = (nflag * 512u + 15u) & ~15u;
    if (TBL_LDS) {
        for (uint32_t i = lane; i < nflag * 256u; i += 64) reinterpret_cast<uint16_t *>(zdw_lds)[i] = tbl[i];
        __builtin_amdgcn_wave_barrier();
    }
    const uint32_t k = blockIdx.x;
    if (k >= nchunks) return;
    const uint64_t s0 = (uint64_t)k << chunk_shift, s1 = s0 + (1ull << chunk_shift);
    uint64_t p;
    if (MODE == 0) p = (k == 0 || s0 <= warm) ? 0 : s0 - warm;
    else p = k == 0 ? 0 : entry_of[k - 1];
    uint32_t n = 0;
    const uint64_t base = MODE == 2 ? rowbase[k] : 0;
    while (p < s1 && p <= limit) {                           // (the section's end is a start too; nothing starts beyond it)
        if (MODE == 2) {
            const uint64_t idx = base + n;
            if (idx > max_rows) break;                       // (the starts of rows 0 .. max_rows: the last one is the section's end)
            if (lane == 0) rstart[idx] = p;
        }
        if (MODE != 0 || p >= s0) ++n;
        p += dec_row_len<TBL_LDS>(w, p, nflag, tbl, lane);
    }
    if (MODE != 2 && lane == 0) {
        exit_out[k] = p;
        if (MODE == 1) count_out[k] = n;
    }
}

// *flag |= 1 when a chunk that lies before the section's end left somewhere else than assumed; entry := exit
__global__ __launch_bounds__(256) void k_dec_check(uint64_t *__restrict__ entry_of, const uint64_t *__restrict__ exit_true, const uint64_t *__restrict__ cum_excl,
                                                   uint32_t nchunks, uint64_t num_rows, uint32_t *__restrict__ flag)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nchunks) return;
    const uint64_t a = entry_of[k], b = exit_true[k];
    if (a != b) {
        entry_of[k] = b;
        if (cum_excl[k] <= num_rows) atomicOr(flag, 1u);
    }
}

// ---- values ----
__device__ __forceinline__ uint64_t load_le(const uint8_t *p, uint32_t size)
{
    // (the buffers carry 8 bytes of slack: an 8-byte read at any position of theirs is inside the allocation)
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return size >= 8 ? v : (v & ((1ull << (8u * size)) - 1ull));
}

__device__ __forceinline__ uint32_t dec_class(uint32_t t);
__device__ __forceinline__ uint32_t dec_strlen(const uint8_t *__restrict__ s, uint64_t room);
// A dictionary-backed value in the matrix: offset of the string (40 bits) | its length + 1 (24 bits; 0 = not measured).  The string
// is measured ONCE, here, for the row that stores the value -- every row that repeats it, and both text kernels, read the length.
constexpr uint64_t DEC_OFF_MASK = (1ull << 40) - 1ull;
__device__ __forceinline__ uint64_t dec_pack_text(uint64_t off, const uint8_t *__restrict__ dict, uint64_t dsize)
{
    if (off == 0 || off >= dsize) return off;                 // empty; or beyond the dictionary (corrupt: printed as empty)
    const uint32_t n = dec_strlen(dict + off, dsize - off);
    return off | ((uint64_t)(n < 0xfffffeu ? n + 1u : 0u) << 40);
}

// wave = row: the values of its flagged columns into V[row][used column]
__global__ __launch_bounds__(256) void k_dec_expand(const uint8_t *__restrict__ rows, const uint64_t *__restrict__ rstart, uint64_t nrows, uint32_t nflag,
                                                    uint32_t nused, const uint8_t *__restrict__ usize, const uint8_t *__restrict__ utype,
                                                    const uint64_t *__restrict__ ubase, const uint8_t *__restrict__ dict, uint64_t dsize,
                                                    uint64_t *__restrict__ V)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < nrows; r += nwaves) {
        const uint64_t p = rstart[r];
        uint32_t run = nflag;
        for (uint32_t ub = 0; ub < nused; ub += 64) {
            const uint32_t u = ub + lane;
            uint32_t sz = 0;
            if (u < nused && ((rows[p + (u >> 3)] >> (u & 7u)) & 1u)) sz = usize[u];
            uint32_t total;
            const uint32_t off = wave_excl_sum(sz, lane, total);
            if (sz) {
                uint64_t v = load_le(rows + p + run + off, sz);
                if (dec_class(utype[u]) <= 1u) v = v ? dec_pack_text(v + ubase[u], dict, dsize) : 0ull;
                V[r * nused + u] = v;
            }
            run += total;
        }
    }
}

constexpr uint32_t DEC_TILE = 64;       // rows per tile

// wave = (tile, batch of 64 used columns): the last flagged value of each column within the tile
__global__ __launch_bounds__(256) void k_dec_tile_last(const uint8_t *__restrict__ rows, const uint64_t *__restrict__ rstart, uint64_t nrows, uint32_t nused,
                                                       uint32_t nbatch, const uint64_t *__restrict__ V, uint64_t *__restrict__ last, uint8_t *__restrict__ has,
                                                       uint64_t ntiles)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t w = wave; w < ntiles * nbatch; w += nwaves) {
        const uint64_t tile = w / nbatch;
        const uint32_t u = (uint32_t)(w % nbatch) * 64u + lane;
        const uint64_t r0 = tile * DEC_TILE, r1 = r0 + DEC_TILE < nrows ? r0 + DEC_TILE : nrows;
        uint64_t val = 0;
        bool any = false;
        if (u < nused) {
            for (uint64_t r = r0; r < r1; ++r) {
                if ((rows[rstart[r] + (u >> 3)] >> (u & 7u)) & 1u) { val = V[r * nused + u]; any = true; }
            }
            last[tile * nused + u] = val;
            has[tile * nused + u] = any ? 1 : 0;
        }
    }
}

constexpr uint32_t DEC_SUPER = 64;      // tiles per superblock of the carry passes

// pass 1: thread = (superblock, column): its last flagged value
__global__ __launch_bounds__(256) void k_dec_carry1(const uint64_t *__restrict__ last, const uint8_t *__restrict__ has, uint64_t ntiles, uint32_t nused,
                                                    uint64_t nsuper, uint64_t *__restrict__ slast, uint8_t *__restrict__ shas)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsuper * nused) return;
    const uint64_t sb = i / nused;
    const uint32_t u = (uint32_t)(i % nused);
    const uint64_t t0 = sb * DEC_SUPER, t1 = t0 + DEC_SUPER < ntiles ? t0 + DEC_SUPER : ntiles;
    uint64_t val = 0;
    bool any = false;
    for (uint64_t t = t0; t < t1; ++t)
        if (has[t * nused + u]) { val = last[t * nused + u]; any = true; }
    slast[i] = val;
    shas[i] = any ? 1 : 0;
}
// pass 2: thread = column: the value before each superblock (in place: slast becomes the carry-in)
__global__ __launch_bounds__(256) void k_dec_carry2(uint64_t *__restrict__ slast, const uint8_t *__restrict__ shas, uint64_t nsuper, uint32_t nused)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nused) return;
    uint64_t val = 0;                                     // 0 before a block's first row
    for (uint64_t sb = 0; sb < nsuper; ++sb) {
        const uint64_t mine = slast[sb * nused + u];
        const bool any = shas[sb * nused + u] != 0;
        slast[sb * nused + u] = val;
        if (any) val = mine;
    }
}
// pass 3: thread = (superblock, column): the value before each of its tiles (in place: last becomes the carry-in)
__global__ __launch_bounds__(256) void k_dec_carry3(uint64_t *__restrict__ last, const uint8_t *__restrict__ has, uint64_t ntiles, uint32_t nused, uint64_t nsuper,
                                                    const uint64_t *__restrict__ scarry)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nsuper * nused) return;
    const uint64_t sb = i / nused;
    const uint32_t u = (uint32_t)(i % nused);
    const uint64_t t0 = sb * DEC_SUPER, t1 = t0 + DEC_SUPER < ntiles ? t0 + DEC_SUPER : ntiles;
    uint64_t val = scarry[i];
    for (uint64_t t = t0; t < t1; ++t) {
        const uint64_t mine = last[t * nused + u];
        const bool any = has[t * nused + u] != 0;
        last[t * nused + u] = val;
        if (any) val = mine;
    }
}

// ---- text ----
enum : uint8_t {   // cplusplus/zdw_column_type_constants.h:17-35
    DT_VARCHAR = 0, DT_TEXT = 1, DT_DATETIME = 2, DT_CHAR_2 = 3, DT_CHAR = 6, DT_TINY = 7, DT_SHORT = 8, DT_LONG = 9, DT_LONGLONG = 10,
    DT_DECIMAL = 11, DT_TINY_SIGNED = 12, DT_SHORT_SIGNED = 13, DT_LONG_SIGNED = 14, DT_LONGLONG_SIGNED = 15, DT_TINYTEXT = 16,
    DT_MEDIUMTEXT = 17, DT_LONGTEXT = 18
};
// 0 text-like (dictionary), 1 decimal (dictionary, default 0.000000000000), 2 char(1), 3 unsigned, 4 signed
__device__ __forceinline__ uint32_t dec_class(uint32_t t)
{
    if (t == DT_DECIMAL) return 1;
    if (t == DT_CHAR) return 2;
    if (t >= DT_TINY && t <= DT_LONGLONG) return 3;
    if (t >= DT_TINY_SIGNED && t <= DT_LONGLONG_SIGNED) return 4;
    return 0;
}
__device__ __forceinline__ uint32_t dec_digits(uint64_t v)
{
    const uint32_t bits = 64u - (uint32_t)__clzll((long long)(v | 1ull));
    const uint32_t t = (bits * 1233u) >> 12;              // floor(log10(2^bits)) or one less
    return v == 0 ? 1u : t + (v >= DEC_POW10[t] ? 1u : 0u);
}
__device__ __forceinline__ uint32_t dec_strlen(const uint8_t *__restrict__ s, uint64_t room)
{
    uint32_t n = 0;
    for (;;) {
        uint64_t w;
        __builtin_memcpy(&w, s + n, 8);
        const uint64_t z = (w - 0x0101010101010101ull) & ~w & 0x8080808080808080ull;      // (the lowest set flag is exact)
        if (z) { n += (uint32_t)(__builtin_ctzll(z) >> 3); break; }
        n += 8;
        if (n >= room) { n = (uint32_t)room; break; }
    }
    return n;
}
// A field's text is put together in registers (three 64-bit words in memory order: numbers have at most 20 digits and a sign,
// + the separator) and leaves in pieces of 8 / 4 / 2 / 1 bytes that add up to exactly its length: the bytes next to it belong to
// other lanes.
struct Text24 { uint64_t w0, w1, w2; };
__device__ __forceinline__ void t24_push_front(Text24 &t, uint32_t c)
{
    t.w2 = (t.w2 << 8) | (t.w1 >> 56);
    t.w1 = (t.w1 << 8) | (t.w0 >> 56);
    t.w0 = (t.w0 << 8) | c;
}
__device__ __forceinline__ void t24_set(Text24 &t, uint32_t at, uint32_t c)          // byte `at` (0..23) := c (it is zero before)
{
    const uint64_t v = (uint64_t)c << (8u * (at & 7u));
    if (at < 8u) t.w0 |= v; else if (at < 16u) t.w1 |= v; else t.w2 |= v;
}
__device__ __forceinline__ void store_exact(uint8_t *dst, const Text24 &t, uint32_t n)
{
    uint32_t o = 0;
    uint64_t w = t.w0;
    if (n >= 8u) { __builtin_memcpy(dst, &t.w0, 8); o = 8; w = t.w1; if (n >= 16u) { __builtin_memcpy(dst + 8, &t.w1, 8); o = 16; w = t.w2; } }
    const uint32_t r = n - o;
    if (r & 4u) { const uint32_t x = (uint32_t)w; __builtin_memcpy(dst + o, &x, 4); w >>= 32; o += 4; }
    if (r & 2u) { const uint16_t x = (uint16_t)w; __builtin_memcpy(dst + o, &x, 2); w >>= 16; o += 2; }
    if (r & 1u) dst[o] = (uint8_t)w;
}
// Decimal digits without a loop: four digits at a time by multiplications that are exact on the ranges they see (24-bit ones: the
// full-rate multiplier), eight ASCII digits per 64-bit word with the most significant in byte 0, i.e. in text order.
__device__ __forceinline__ uint32_t dec_ascii4(uint32_t v)          // v < 10000: "dddd" (leading zeros kept)
{
    const uint32_t q100 = __umul24(v, 10486u) >> 20;                 // v / 100 (exact below 43668)
    const uint32_t r100 = v - __umul24(q100, 100u);
    const uint32_t w = q100 | (r100 << 16);                          // two numbers below 100 in 16-bit halves
    const uint32_t q10 = (__umul24(w, 103u) >> 10) & 0x000f000fu;    // each / 10 (exact below 179; the halves do not meet)
    const uint32_t r10 = w - __umul24(q10, 10u);
    return (q10 | (r10 << 8)) + 0x30303030u;
}
__device__ __forceinline__ uint64_t dec_ascii8(uint32_t v)          // v < 10^8
{
    const uint32_t hi = __umulhi(v, 0xD1B71759u) >> 13;              // v / 10000
    const uint32_t lo = v - hi * 10000u;
    return (uint64_t)dec_ascii4(hi) | ((uint64_t)dec_ascii4(lo) << 32);
}
// t |= the 8 bytes of d at byte position pos (0..16)
__device__ __forceinline__ void t24_or_at(Text24 &t, uint32_t pos, uint64_t d)
{
    const uint32_t sh = 8u * (pos & 7u);
    const uint64_t lo = d << sh, hi = sh ? d >> (64u - sh) : 0ull;
    const uint32_t idx = pos >> 3;                      // (selects, not branches: an indexed struct would live in scratch memory)
    t.w0 |= idx == 0u ? lo : 0ull;
    t.w1 |= idx == 0u ? hi : (idx == 1u ? lo : 0ull);
    t.w2 |= idx == 1u ? hi : (idx == 2u ? lo : 0ull);
}
// the nd = dec_digits(v) decimal digits of v at the front of t (which is empty)
__device__ __forceinline__ void t24_digits(Text24 &t, uint64_t v, uint32_t nd)
{
    if (v < 100000000ull) {                                          // nd <= 8
        t.w0 = dec_ascii8((uint32_t)v) >> (8u * (8u - nd));
        return;
    }
    if (!(v >> 32)) {                                                // 9 or 10 digits: all in 32-bit arithmetic
        const uint32_t x = (uint32_t)v;
        const uint32_t qh = __umulhi(x >> 8, 0xABCC7712u) >> 18;     // x / 10^8 (10^8 = 2^8 * 390625; exact below 2^32), at most 42
        const uint32_t k = nd - 8u;
        t.w0 = (uint64_t)(dec_ascii4(qh) >> (8u * (4u - k)));
        t24_or_at(t, k, dec_ascii8(x - qh * 100000000u));
        return;
    }
    const uint64_t q = v / 100000000ull;
    const uint64_t d_low = dec_ascii8((uint32_t)(v - q * 100000000ull));
    if (q < 100000000ull) {                                          // up to 16 digits: k of them in front of the low eight
        const uint32_t k = nd - 8u;
        t.w0 = dec_ascii8((uint32_t)q) >> (8u * (8u - k));
        t24_or_at(t, k, d_low);
        return;
    }
    const uint32_t q2 = (uint32_t)(q / 100000000ull);                // 17..20 digits: below 10^4
    const uint64_t d_mid = dec_ascii8((uint32_t)(q - (uint64_t)q2 * 100000000ull));
    const uint32_t k = nd - 16u;
    t.w0 = (uint64_t)(dec_ascii4(q2) >> (8u * (4u - k)));
    t24_or_at(t, k, d_mid);
    t24_or_at(t, k + 8u, d_low);
}
// n bytes from src, then the separator
__device__ inline void dec_copy_sep(uint8_t *dst, const uint8_t *src, uint32_t n, uint32_t sep)
{
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; __builtin_memcpy(&w, src + i, 8); __builtin_memcpy(dst + i, &w, 8); }
    uint64_t w;
    __builtin_memcpy(&w, src + i, 8);                // (the dictionary carries 8 bytes of slack)
    const uint32_t r = n - i;                        // < 8
    w = (r ? (w & ((1ull << (8u * r)) - 1ull)) : 0ull) | ((uint64_t)sep << (8u * r));
    Text24 t{w, 0, 0};
    store_exact(dst + i, t, r + 1u);
}

// The text of one field and the separator behind it: its length (separator included); EMIT: written to dst.  `used`: the column has
// values in this block (else: its default, outputDefault, UnconvertFromZDW.cpp:1224-1267); val: the stored value after repeat-fill.
template <bool EMIT>
__device__ __forceinline__ uint32_t dec_field(uint32_t type, bool used, uint64_t val, uint64_t base, const uint8_t *__restrict__ dict, uint64_t dsize,
                                              uint8_t *dst, uint32_t sep)
{
    const uint32_t cls = dec_class(type);
    Text24 t{0, 0, 0};
    uint32_t n = 0;
    if (cls <= 1) {
        if (used && val) {
            const uint64_t off = val & DEC_OFF_MASK;                                 // (packed by k_dec_expand: absolute, with its length)
            if (off < dsize) {                                                       // (beyond it: corrupt data, printed as empty)
                n = (val >> 40) ? (uint32_t)(val >> 40) - 1u : dec_strlen(dict + off, dsize - off);
                if (EMIT) dec_copy_sep(dst, dict + off, n, sep);
                return n + 1u;
            }
        } else if (cls == 1) {
            t.w0 = 0x303030302e30ull | (0x3030ull << 48);                            // "0.000000" ...
            t.w1 = 0x303030303030ull;                                                // ... "000000"
            n = 14;
        }
    } else if (cls == 2) {                                                           // :1396-1420
        if (used && val) {
            const uint64_t tuple = val + base;
            const uint32_t b0 = (uint32_t)tuple & 0xffu, b1 = (uint32_t)(tuple >> 8) & 0xffu;
            if (b0 != '\\') { if (b0) { t.w0 = b0; n = 1; } }
            else { t.w0 = b0 | (b1 << 8); n = 2; }
        }
    } else {
        uint64_t x = (used && val) ? val + base : 0;
        uint32_t neg = 0;
        if (cls == 4 && (int64_t)x < 0) { neg = 1; x = 0ull - x; }
        const uint32_t nd = dec_digits(x);
        n = neg + nd;
        if (EMIT) { t24_digits(t, x, nd); if (neg) t24_push_front(t, '-'); }
    }
    if (EMIT) { t24_set(t, n, sep); store_exact(dst, t, n + 1u); }
    return n + 1u;
}

struct DecText {
    const uint8_t *rows; const uint64_t *rstart; uint64_t nrows;
    uint32_t ncols, nused;
    const uint8_t *type;        // [ncols]
    const int32_t *used_of;     // [ncols]: index among the used columns, -1: unused in this block
    const uint64_t *ubase;      // [nused]
    const uint64_t *V, *carry;  // [nrows][nused], [ntiles][nused]
    const uint8_t *dict; uint64_t dsize;
    uint32_t *rowlen;           // [nrows] (LEN)
    uint32_t *pre;              // [nrows][batches of 64 columns]: where the batch begins within its row's text (LEN writes, EMIT reads)
    const uint64_t *rowoff;     // [nrows] (EMIT)
    uint8_t *out;
};

// wave = tile of 64 rows; lane = column (batches of 64 schema columns), walking the tile's rows with the value so far in a
// register.  LEN: the rows' text lengths (lane i keeps row i's) and where each batch begins within its row.  EMIT: the text.
template <bool EMIT>
__global__ __launch_bounds__(256) void k_dec_text(const DecText a, uint64_t ntiles, uint32_t nbatch)
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t tile = wave; tile < ntiles; tile += nwaves) {
        const uint64_t r0 = tile * DEC_TILE, r1 = r0 + DEC_TILE < a.nrows ? r0 + DEC_TILE : a.nrows;
        uint32_t mylen = 0;                                    // lane i: text bytes of row r0 + i so far
        for (uint32_t b = 0; b < nbatch; ++b) {
            const uint32_t c = b * 64u + lane;
            const bool col = c < a.ncols;
            const int32_t u = col ? a.used_of[c] : -1;
            const uint32_t type = col ? a.type[c] : 0u;
            const uint64_t base = u >= 0 ? a.ubase[u] : 0ull;
            const uint32_t sep = c + 1u == a.ncols ? '\n' : '\t';
            uint64_t val = u >= 0 ? a.carry[tile * a.nused + (uint32_t)u] : 0ull;
            // (the next row's flag byte and value are requested while this one is worked on; a value is read whether flagged or not)
            uint32_t fcur = u >= 0 ? a.rows[a.rstart[r0] + ((uint32_t)u >> 3)] : 0u, fnext = 0;
            uint64_t vcur = u >= 0 ? a.V[r0 * a.nused + (uint32_t)u] : 0ull, vnext = 0;
            for (uint64_t r = r0; r < r1; ++r) {
                if (u >= 0 && r + 1 < r1) { fnext = a.rows[a.rstart[r + 1] + ((uint32_t)u >> 3)]; vnext = a.V[(r + 1) * a.nused + (uint32_t)u]; }
                if (u >= 0 && ((fcur >> ((uint32_t)u & 7u)) & 1u)) val = vcur;
                fcur = fnext; vcur = vnext;
                const uint32_t ri = (uint32_t)(r - r0);
                const uint32_t len = col ? dec_field<false>(type, u >= 0, val, base, a.dict, a.dsize, nullptr, sep) : 0u;
                uint32_t total;
                const uint32_t off = wave_excl_sum(len, lane, total);
                if (!EMIT) {
                    if (lane == ri) { a.pre[r * nbatch + b] = mylen; mylen += total; }
                } else if (col) {
                    (void)dec_field<true>(type, u >= 0, val, base, a.dict, a.dsize, a.out + a.rowoff[r] + a.pre[r * nbatch + b] + off, sep);
                }
            }
        }
        if (!EMIT && r0 + lane < r1) a.rowlen[r0 + lane] = mylen;
    }
}

// The text of schemas of up to 512 columns whose rows fit DEC_ROWBUF bytes: a wave walks the 64 rows of its tile one by one with
// every batch's value so far in registers, puts the row together in LDS (lane = column; where a field goes: a prefix sum of the
// lengths) and copies it out with 16-byte stores.  (The kernel above writes every field straight to memory in pieces of 8 / 4 / 2 /
// 1 bytes: three to five store instructions per field against one LDS write and a sixteenth of a global store.)
constexpr uint32_t DEC_ROWBUF = 8192;

// a field ready to be written: its length, and either its text in registers (<= 21 bytes + separator) or where its bytes lie
struct DecField { uint32_t len; bool from_dict; const uint8_t *src; Text24 t; };
__device__ __forceinline__ DecField dec_prepare(uint32_t type, bool used, uint64_t val, uint64_t base, const uint8_t *__restrict__ dict, uint64_t dsize)
{
    DecField f;
    f.len = 0; f.from_dict = false; f.src = nullptr; f.t = Text24{0, 0, 0};
    const uint32_t cls = dec_class(type);
    if (cls <= 1) {
        if (used && val) {
            const uint64_t off = val & DEC_OFF_MASK;
            if (off < dsize) { f.len = (val >> 40) ? (uint32_t)(val >> 40) - 1u : dec_strlen(dict + off, dsize - off); f.from_dict = true; f.src = dict + off; }
        } else if (cls == 1) {
            f.t.w0 = 0x303030302e30ull | (0x3030ull << 48); f.t.w1 = 0x303030303030ull; f.len = 14;
        }
    } else if (cls == 2) {
        if (used && val) {
            const uint64_t tuple = val + base;
            const uint32_t b0 = (uint32_t)tuple & 0xffu, b1 = (uint32_t)(tuple >> 8) & 0xffu;
            if (b0 != '\\') { if (b0) { f.t.w0 = b0; f.len = 1; } }
            else { f.t.w0 = b0 | (b1 << 8); f.len = 2; }
        }
    } else {
        uint64_t x = (used && val) ? val + base : 0;
        uint32_t neg = 0;
        if (cls == 4 && (int64_t)x < 0) { neg = 1; x = 0ull - x; }
        const uint32_t nd = dec_digits(x);
        f.len = neg + nd;
        t24_digits(f.t, x, nd);
        if (neg) t24_push_front(f.t, '-');
    }
    return f;
}
// into LDS at byte offset `at`: the field and its separator
__device__ __forceinline__ void dec_put_lds(uint32_t at, DecField &f, uint32_t sep)
{
    uint8_t *dst = zdw_lds + at;
    if (f.from_dict) {
        uint32_t i = 0;
        for (; i + 8 <= f.len; i += 8) { uint64_t w; __builtin_memcpy(&w, f.src + i, 8); __builtin_memcpy(dst + i, &w, 8); }
        uint64_t w;
        __builtin_memcpy(&w, f.src + i, 8);
        const uint32_t r = f.len - i;
        w = (r ? (w & ((1ull << (8u * r)) - 1ull)) : 0ull) | ((uint64_t)sep << (8u * r));
        Text24 t{w, 0, 0};
        store_exact(dst + i, t, r + 1u);
    } else {
        t24_set(f.t, f.len, sep);
        store_exact(dst, f.t, f.len + 1u);
    }
}

// A wave's unit of work: a tile of 64 rows x a group of NB batches of 64 columns (the whole row when the schema has no more).  A
// group's piece of a row begins where the lengths kernel said (pre).
template <uint32_t NB>
__global__ __launch_bounds__(256) void k_dec_rows(const DecText a, uint64_t ntiles, uint32_t nbatch, uint32_t ngroups)
{
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint32_t buf = wv * DEC_ROWBUF;                       // this wave's row buffer in LDS
    for (uint64_t unit = wave; unit < ntiles * ngroups; unit += nwaves) {
        const uint64_t tile = unit / ngroups;
        const uint32_t b0 = (uint32_t)(unit - tile * ngroups) * NB;          // the group's first batch
        const bool last_group = b0 + NB >= nbatch;
        // the wave's columns
        bool col[NB]; int32_t u[NB]; uint32_t type[NB], sep[NB]; uint64_t base[NB];
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) {
            const uint32_t c = (b0 + b) * 64u + lane;
            col[b] = c < a.ncols;
            u[b] = col[b] ? a.used_of[c] : -1;
            type[b] = col[b] ? a.type[c] : 0u;
            base[b] = u[b] >= 0 ? a.ubase[u[b]] : 0ull;
            sep[b] = c + 1u == a.ncols ? '\n' : '\t';
        }
        const uint64_t r0 = tile * DEC_TILE, r1 = r0 + DEC_TILE < a.nrows ? r0 + DEC_TILE : a.nrows;
        uint64_t val[NB];
#pragma unroll
        for (uint32_t b = 0; b < NB; ++b) val[b] = u[b] >= 0 ? a.carry[tile * a.nused + (uint32_t)u[b]] : 0ull;
        // a row's flag bytes and values are requested one row ahead (a value is read whether its flag is set or not: V has a cell
        // for every used column of every row)
        uint32_t fb[NB], fb_next[NB];
        uint64_t vv[NB], vv_next[NB];
        auto request = [&](uint64_t r, uint32_t (&f)[NB], uint64_t (&v)[NB]) {
            const uint64_t p = a.rstart[r];
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                f[b] = u[b] >= 0 ? a.rows[p + ((uint32_t)u[b] >> 3)] : 0u;
                v[b] = u[b] >= 0 ? a.V[r * a.nused + (uint32_t)u[b]] : 0ull;
            }
        };
        request(r0, fb, vv);
        for (uint64_t r = r0; r < r1; ++r) {
            // this group's piece of the row: [o0, o0 + seglen) of the text
            const uint64_t row0 = a.rowoff[r];
            const uint32_t at0 = a.pre[r * nbatch + b0];
            const uint32_t at1 = last_group ? (uint32_t)(a.rowoff[r + 1] - row0) : a.pre[r * nbatch + b0 + NB];
            const uint64_t o0 = row0 + at0;
            const uint32_t seglen = at1 - at0;
            const bool in_lds = seglen <= DEC_ROWBUF;
            if (r + 1 < r1) request(r + 1, fb_next, vv_next);
            uint32_t run = 0;
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) {
                if (u[b] >= 0 && ((fb[b] >> ((uint32_t)u[b] & 7u)) & 1u)) val[b] = vv[b];
                DecField f = dec_prepare(type[b], u[b] >= 0, val[b], base[b], a.dict, a.dsize);
                const uint32_t len = col[b] ? f.len + 1u : 0u;
                uint32_t total;
                const uint32_t off = wave_excl_sum(len, lane, total);
                if (col[b]) {
                    if (in_lds) dec_put_lds(buf + run + off, f, sep[b]);
                    else (void)dec_field<true>(type[b], u[b] >= 0, val[b], base[b], a.dict, a.dsize, a.out + o0 + run + off, sep[b]);      // a piece longer than the buffer
                }
                run += total;
            }
            if (in_lds) {
                __builtin_amdgcn_wave_barrier();
                // out[o0, o0 + seglen) <- buffer: whole 16-byte pieces, then what is left byte by byte
                uint8_t *dst = a.out + o0;
                const uint32_t whole = seglen & ~15u;
                for (uint32_t k = lane * 16u; k < whole; k += 64u * 16u) {
                    const uint4 w = *reinterpret_cast<const uint4 *>(zdw_lds + buf + k);
                    __builtin_memcpy(dst + k, &w, 16);
                }
                if (lane < (seglen & 15u)) dst[whole + lane] = zdw_lds[buf + whole + lane];
                __builtin_amdgcn_wave_barrier();
            }
#pragma unroll
            for (uint32_t b = 0; b < NB; ++b) { fb[b] = fb_next[b]; vv[b] = vv_next[b]; }
        }
    }
}

// ------------------------------------------------------------------------------------------
// host side: the zdw_dec_* C ABI
// ------------------------------------------------------------------------------------------
enum : int { DEC_OK = 0, DEC_UNSUPPORTED_VERSION = 5, DEC_TOO_SHORT = 6, DEC_BAD_TYPE = 7, DEC_CORRUPT = 9, DEC_DEVICE = 20 };

struct DBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 256 ? 256 : bytes + bytes / 16;
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want; else { p = nullptr; (void)hipGetLastError(); }
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

std::string g_dec_create_error;

}  // namespace
}  // namespace zdw

using namespace zdw;

struct zdw_dec {
    int device = 0;
    hipStream_t s = nullptr;
    std::string err;
    // the stream
    DBuf img;
    uint64_t len = 0, pos = 0;
    bool have_input = false, ended = false;
    uint32_t version = 0;
    std::vector<std::string> names;
    std::vector<uint8_t> types;
    std::vector<uint16_t> charsizes;
    // the block
    DBuf d_type, d_used_of, d_usize, d_utype, d_ubase, d_tbl, entry, exit_true, counts, cum, scan_tmp, rstart, V, last, has, slast, shas, rowlen, rowoff, pre, text, flag;
    uint64_t text_bytes = 0;
    bool have_block = false;
    float stage_ms[5] = {0, 0, 0, 0, 0};
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    // pinned pieces for zdw_dec_text_write
    void *piece[2] = {nullptr, nullptr};
    hipEvent_t piece_ev[2] = {nullptr, nullptr};
};

namespace {

int dfail(zdw_dec *d, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    d->err = buf;
    return code;
}
#define DEC_TRY(d, expr)                                                                                    \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) { (void)hipGetLastError(); return dfail(d, DEC_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } \
    } while (0)

constexpr uint64_t PIECE = 32ull << 20;

}  // namespace

extern "C" {

int zdw_dec_create(int device, zdw_dec **out)
{
    if (!out) return DEC_DEVICE;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { g_dec_create_error = "no HIP device"; return DEC_DEVICE; }
    if (device < 0) device = 0;
    if (device >= count) { g_dec_create_error = "no such device"; return DEC_DEVICE; }
    zdw_dec *d = new (std::nothrow) zdw_dec;
    if (!d) { g_dec_create_error = "out of host memory"; return DEC_DEVICE; }
    d->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&d->s, hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < 6; ++i) e = hipEventCreate(&d->ev[i]);
    if (e != hipSuccess) { g_dec_create_error = hipGetErrorString(e); zdw_dec_destroy(d); return DEC_DEVICE; }
    *out = d;
    return DEC_OK;
}

void zdw_dec_destroy(zdw_dec *d)
{
    if (!d) return;
    (void)hipSetDevice(d->device);
    if (d->s) (void)hipStreamSynchronize(d->s);
    for (DBuf *b : {&d->img, &d->d_type, &d->d_used_of, &d->d_usize, &d->d_utype, &d->d_ubase, &d->d_tbl, &d->entry, &d->exit_true, &d->counts, &d->cum, &d->scan_tmp,
                    &d->rstart, &d->V, &d->last, &d->has, &d->slast, &d->shas, &d->rowlen, &d->rowoff, &d->pre, &d->text, &d->flag})
        b->release();
    for (int i = 0; i < 6; ++i) if (d->ev[i]) (void)hipEventDestroy(d->ev[i]);
    for (int i = 0; i < 2; ++i) { if (d->piece[i]) (void)hipHostFree(d->piece[i]); if (d->piece_ev[i]) (void)hipEventDestroy(d->piece_ev[i]); }
    if (d->s) (void)hipStreamDestroy(d->s);
    delete d;
}

const char *zdw_dec_last_error(const zdw_dec *d) { return d ? d->err.c_str() : g_dec_create_error.c_str(); }

int zdw_dec_set_input_host(zdw_dec *d, const uint8_t *zdw, uint64_t len)
{
    if (!d) return DEC_DEVICE;
    if (len && !zdw) return dfail(d, DEC_DEVICE, "zdw_dec_set_input_host: zdw is NULL");
    (void)hipSetDevice(d->device);
    d->have_input = d->have_block = d->ended = false;
    d->names.clear(); d->types.clear(); d->charsizes.clear();
    // file header (ConvertToZDW.cpp:673-737 writes it, UnconvertFromZDW.cpp:758-860 reads it): host memory is at hand here
    const uint8_t *p = zdw, *end = zdw + len;
    auto need = [&](uint64_t n) { return (uint64_t)(end - p) >= n; };
    if (!need(2)) return dfail(d, DEC_CORRUPT, "no version field");
    d->version = (uint32_t)p[0] | ((uint32_t)p[1] << 8);
    p += 2;
    if (d->version < 9 || d->version > 11) return dfail(d, DEC_UNSUPPORTED_VERSION, "ZDW version %u is not supported (9-11)", d->version);
    if (d->version >= 11) {
        if (!need(4)) return dfail(d, DEC_CORRUPT, "metadata length missing");
        const uint64_t mlen = (uint64_t)p[0] | ((uint64_t)p[1] << 8) | ((uint64_t)p[2] << 16) | ((uint64_t)p[3] << 24);
        p += 4;
        if (!need(mlen)) return dfail(d, DEC_CORRUPT, "metadata block cut short");
        p += mlen;
    }
    for (;;) {
        const uint8_t *z = static_cast<const uint8_t *>(memchr(p, 0, (size_t)(end - p)));
        if (!z) return dfail(d, DEC_CORRUPT, "column names cut short");
        if (z == p) { ++p; break; }
        d->names.emplace_back(reinterpret_cast<const char *>(p), (size_t)(z - p));
        p = z + 1;
    }
    const size_t n = d->names.size();
    if (!need(3 * n)) return dfail(d, DEC_CORRUPT, "column types cut short");
    d->types.assign(p, p + n);
    p += n;
    d->charsizes.resize(n);
    for (size_t c = 0; c < n; ++c) { d->charsizes[c] = (uint16_t)(p[0] | (p[1] << 8)); p += 2; }
    for (size_t c = 0; c < n; ++c) {
        const uint8_t t = d->types[c];
        const bool known = t <= 3 || (t >= 6 && t <= 18);
        if (!known) return dfail(d, DEC_BAD_TYPE, "column %zu has type %u", c, (unsigned)t);
    }
    d->pos = (uint64_t)(p - zdw);
    d->len = len;
    DEC_TRY(d, d->img.ensure(len + 64));
    if (len) DEC_TRY(d, hipMemcpyAsync(d->img.p, zdw, len, hipMemcpyHostToDevice, d->s));
    DEC_TRY(d, hipMemsetAsync(d->img.as<uint8_t>() + len, 0, 64, d->s));
    DEC_TRY(d, d->d_type.ensure(n + 8));
    if (n) DEC_TRY(d, hipMemcpyAsync(d->d_type.p, d->types.data(), n, hipMemcpyHostToDevice, d->s));
    DEC_TRY(d, hipStreamSynchronize(d->s));
    d->ended = d->pos == len;                    // an empty input gives a header without blocks (ConvertToZDW.cpp:824-836)
    d->have_input = true;
    return DEC_OK;
}

int zdw_dec_header(const zdw_dec *d, uint32_t *version, uint32_t *ncols)
{
    if (!d || !d->have_input) return DEC_DEVICE;
    if (version) *version = d->version;
    if (ncols) *ncols = (uint32_t)d->names.size();
    return DEC_OK;
}

int zdw_dec_column(const zdw_dec *d, uint32_t i, const char **name, uint8_t *type, uint16_t *char_size)
{
    if (!d || !d->have_input || i >= d->names.size()) return DEC_DEVICE;
    if (name) *name = d->names[i].c_str();
    if (type) *type = d->types[i];
    if (char_size) *char_size = d->charsizes[i];
    return DEC_OK;
}

int zdw_dec_next_block(zdw_dec *d, zdw_dec_block *blk)
{
    if (!d || !blk) return DEC_DEVICE;
    memset(blk, 0, sizeof *blk);
    if (!d->have_input) return dfail(d, DEC_DEVICE, "no input set");
    (void)hipSetDevice(d->device);
    d->have_block = false;
    if (d->ended) { blk->is_last_block = 1; return DEC_OK; }
    hipStream_t s = d->s;
    const uint8_t *img = d->img.as<uint8_t>();
    const uint32_t ncols = (uint32_t)d->names.size();
    auto fetch = [&](uint64_t off, void *dst, uint64_t nbytes) -> bool {
        if (nbytes > d->len || off > d->len - nbytes) return false;      // (compared without adding: untrusted sizes must not wrap)
        return hipMemcpy(dst, img + off, nbytes, hipMemcpyDeviceToHost) == hipSuccess;
    };
    uint64_t pos = d->pos;
    uint8_t hdr[9 + 9];
    memset(hdr, 0, sizeof hdr);
    if (!fetch(pos, hdr, 9)) return dfail(d, DEC_CORRUPT, "block header cut short");
    const uint32_t nrows = (uint32_t)hdr[0] | ((uint32_t)hdr[1] << 8) | ((uint32_t)hdr[2] << 16) | ((uint32_t)hdr[3] << 24);
    blk->num_rows = nrows;
    blk->longest_line = (uint32_t)hdr[4] | ((uint32_t)hdr[5] << 8) | ((uint32_t)hdr[6] << 16) | ((uint32_t)hdr[7] << 24);
    blk->is_last_block = hdr[8] != 0;
    pos += 9;
    // dictionary section (dictionary.cpp:76-111): 00, or index size, size, origin byte + strings
    uint8_t isz = 0;
    if (!fetch(pos, &isz, 1)) return dfail(d, DEC_CORRUPT, "dictionary header cut short");
    pos += 1;
    uint64_t dsize = 0, dict_off = 0;
    if (isz) {
        uint8_t sz[8] = {0};
        // (the format's dictionary offsets are 32-bit -- dictionary.cpp:62-73 -- so the size field of a v9-v11 stream has at most 4 bytes
        // and the dictionary fewer than 2^32)
        if (isz > 4 || !fetch(pos, sz, isz)) return dfail(d, DEC_CORRUPT, "dictionary size field");
        for (unsigned i = 0; i < isz; ++i) dsize |= (uint64_t)sz[i] << (8 * i);
        pos += isz;
        uint8_t lastb = 1;
        if (dsize == 0 || dsize >= (1ull << 32) || dsize > d->len - pos || !fetch(pos + dsize - 1, &lastb, 1) || lastb != 0)
            return dfail(d, DEC_CORRUPT, "dictionary cut short");
        dict_off = pos;
        pos += dsize;
    }
    blk->dict_bytes = dsize;
    std::vector<uint8_t> size(ncols ? ncols : 1);
    if (ncols && !fetch(pos, size.data(), ncols)) return dfail(d, DEC_CORRUPT, "column sizes cut short");
    pos += ncols;
    std::vector<int32_t> used_of(ncols ? ncols : 1, -1);
    std::vector<uint8_t> usize, utype;
    for (uint32_t c = 0; c < ncols; ++c) {
        if (size[c] > 8) return dfail(d, DEC_CORRUPT, "column %u is %u bytes wide", c, (unsigned)size[c]);
        if (size[c]) { used_of[c] = (int32_t)usize.size(); usize.push_back(size[c]); utype.push_back(d->types[c]); }
    }
    const uint32_t nused = (uint32_t)usize.size();
    std::vector<uint64_t> ubase(nused ? nused : 1);
    if (nused && !fetch(pos, ubase.data(), 8ull * nused)) return dfail(d, DEC_CORRUPT, "column bases cut short");
    pos += 8ull * nused;
    const uint64_t ro = pos, avail = d->len - ro;
    const uint32_t nflag = (nused + 7) / 8;
    const uint8_t *rows = img + ro;

    DEC_TRY(d, d->d_used_of.ensure(4ull * ncols + 8));
    DEC_TRY(d, d->d_usize.ensure(nused + 8));
    DEC_TRY(d, d->d_utype.ensure(nused + 8));
    DEC_TRY(d, d->d_ubase.ensure(8ull * nused + 8));
    if (ncols) DEC_TRY(d, hipMemcpyAsync(d->d_used_of.p, used_of.data(), 4ull * ncols, hipMemcpyHostToDevice, s));
    if (nused) {
        DEC_TRY(d, hipMemcpyAsync(d->d_usize.p, usize.data(), nused, hipMemcpyHostToDevice, s));
        DEC_TRY(d, hipMemcpyAsync(d->d_utype.p, utype.data(), nused, hipMemcpyHostToDevice, s));
        DEC_TRY(d, hipMemcpyAsync(d->d_ubase.p, ubase.data(), 8ull * nused, hipMemcpyHostToDevice, s));
    }
    DEC_TRY(d, d->rstart.ensure(8ull * ((uint64_t)nrows + 2)));
    DEC_TRY(d, hipEventRecord(d->ev[0], s));
    uint64_t rows_end = 0;
    blk->walk_passes = 0;
    if (nrows && nflag) {
        // ---- where the rows start ----
        uint64_t maxrow = nflag;
        for (uint8_t z : usize) maxrow += z;
        const uint64_t region = std::min<uint64_t>(avail, (uint64_t)nrows * maxrow);      // the section cannot be longer
        const uint32_t chunk_shift = 19;
        const uint64_t nchunks64 = (region >> chunk_shift) + 2;                           // (one past the end: the section's end is a start too)
        if (nchunks64 > 0x7fffffffull) return dfail(d, DEC_DEVICE, "row section too large");
        const uint32_t nchunks = (uint32_t)nchunks64;
        std::vector<uint16_t> tbl((size_t)nflag * 256);
        for (uint32_t j = 0; j < nflag; ++j)
            for (uint32_t b = 0; b < 256; ++b) {
                uint32_t sum = 0;
                for (uint32_t i = 0; i < 8; ++i) if ((b >> i) & 1u) { const uint32_t u = 8 * j + i; if (u < nused) sum += usize[u]; }
                tbl[(size_t)j * 256 + b] = (uint16_t)sum;
            }
        DEC_TRY(d, d->d_tbl.ensure(tbl.size() * 2));
        DEC_TRY(d, hipMemcpyAsync(d->d_tbl.p, tbl.data(), tbl.size() * 2, hipMemcpyHostToDevice, s));
        DEC_TRY(d, d->entry.ensure(8ull * nchunks));
        DEC_TRY(d, d->exit_true.ensure(8ull * nchunks));
        DEC_TRY(d, d->counts.ensure(4ull * nchunks + 8));
        DEC_TRY(d, d->cum.ensure(8ull * (nchunks + 2)));
        DEC_TRY(d, d->scan_tmp.ensure(scan_tmp_bytes(nchunks + 1ull)));
        DEC_TRY(d, d->flag.ensure(64));
        const bool lds = (size_t)nflag * 512 <= 32768;
        const size_t shm = lds ? (((size_t)nflag * 512 + 15) & ~(size_t)15) + 2 * 8192 : 0;
        const uint64_t warm = 1ull << chunk_shift;
        const uint16_t *t16 = d->d_tbl.as<uint16_t>();
        if (lds) hipLaunchKernelGGL((k_dec_walk<0, true>), dim3(nchunks), dim3(64), shm, s, rows, avail, nflag, t16, chunk_shift, nchunks, warm, nullptr,
                                    d->entry.as<uint64_t>(), nullptr, nullptr, nullptr, 0ull);
        else hipLaunchKernelGGL((k_dec_walk<0, false>), dim3(nchunks), dim3(64), 0, s, rows, avail, nflag, t16, chunk_shift, nchunks, warm, nullptr,
                                d->entry.as<uint64_t>(), nullptr, nullptr, nullptr, 0ull);
        for (uint32_t pass = 0;; ++pass) {
            if (lds) hipLaunchKernelGGL((k_dec_walk<1, true>), dim3(nchunks), dim3(64), shm, s, rows, avail, nflag, t16, chunk_shift, nchunks, warm,
                                        d->entry.as<uint64_t>(), d->exit_true.as<uint64_t>(), d->counts.as<uint32_t>(), nullptr, nullptr, 0ull);
            else hipLaunchKernelGGL((k_dec_walk<1, false>), dim3(nchunks), dim3(64), 0, s, rows, avail, nflag, t16, chunk_shift, nchunks, warm,
                                    d->entry.as<uint64_t>(), d->exit_true.as<uint64_t>(), d->counts.as<uint32_t>(), nullptr, nullptr, 0ull);
            launch_scan_u32_to_u64(d->counts.as<uint32_t>(), d->cum.as<uint64_t>(), nchunks, d->scan_tmp.p, nullptr, s);
            DEC_TRY(d, hipMemsetAsync(d->flag.p, 0, 4, s));
            hipLaunchKernelGGL(k_dec_check, dim3((nchunks + 255) / 256), dim3(256), 0, s, d->entry.as<uint64_t>(), d->exit_true.as<uint64_t>(),
                               d->cum.as<uint64_t>(), nchunks, (uint64_t)nrows, d->flag.as<uint32_t>());
            uint32_t wrong = 0;
            DEC_TRY(d, hipMemcpyAsync(&wrong, d->flag.p, 4, hipMemcpyDeviceToHost, s));
            DEC_TRY(d, hipStreamSynchronize(s));
            blk->walk_passes = pass + 1;
            if (!wrong) break;
            if (pass > nchunks + 2) return dfail(d, DEC_DEVICE, "row chain did not settle");
        }
        uint64_t found = 0;
        DEC_TRY(d, hipMemcpy(&found, d->cum.as<uint64_t>() + nchunks, 8, hipMemcpyDeviceToHost));
        if (found <= nrows) return dfail(d, DEC_TOO_SHORT, "the stream ends before the block's %u rows do", nrows);
        if (lds) hipLaunchKernelGGL((k_dec_walk<2, true>), dim3(nchunks), dim3(64), shm, s, rows, avail, nflag, t16, chunk_shift, nchunks, warm,
                                    d->entry.as<uint64_t>(), nullptr, nullptr, d->cum.as<uint64_t>(), d->rstart.as<uint64_t>(), (uint64_t)nrows);
        else hipLaunchKernelGGL((k_dec_walk<2, false>), dim3(nchunks), dim3(64), 0, s, rows, avail, nflag, t16, chunk_shift, nchunks, warm,
                                d->entry.as<uint64_t>(), nullptr, nullptr, d->cum.as<uint64_t>(), d->rstart.as<uint64_t>(), (uint64_t)nrows);
        DEC_TRY(d, hipMemcpyAsync(&rows_end, d->rstart.as<uint64_t>() + nrows, 8, hipMemcpyDeviceToHost, s));
        DEC_TRY(d, hipStreamSynchronize(s));
        if (rows_end > avail) return dfail(d, DEC_TOO_SHORT, "the stream ends before the block's %u rows do", nrows);
    } else {
        DEC_TRY(d, hipMemsetAsync(d->rstart.p, 0, 8ull * ((uint64_t)nrows + 2), s));
    }
    DEC_TRY(d, hipEventRecord(d->ev[1], s));

    // ---- values: flagged ones into the matrix, then what precedes each tile ----
    const uint64_t ntiles = ((uint64_t)nrows + DEC_TILE - 1) / DEC_TILE, nsuper = (ntiles + DEC_SUPER - 1) / DEC_SUPER;
    const uint32_t ubatch = (nused + 63) / 64;
    DEC_TRY(d, d->V.ensure((uint64_t)nrows * nused * 8 + 64));
    DEC_TRY(d, d->last.ensure(ntiles * nused * 8 + 64));
    DEC_TRY(d, d->has.ensure(ntiles * nused + 64));
    DEC_TRY(d, d->slast.ensure(nsuper * nused * 8 + 64));
    DEC_TRY(d, d->shas.ensure(nsuper * nused + 64));
    int cus = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, d->device) == hipSuccess) cus = prop.multiProcessorCount; }
    auto grid_for = [&](uint64_t waves) { const uint64_t wgs = (waves + 3) / 4; return (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(wgs, (uint64_t)cus * 8)); };
    if (nrows && nused) {
        hipLaunchKernelGGL(k_dec_expand, dim3(grid_for(nrows)), dim3(256), 0, s, rows, d->rstart.as<uint64_t>(), (uint64_t)nrows, nflag, nused,
                           d->d_usize.as<uint8_t>(), d->d_utype.as<uint8_t>(), d->d_ubase.as<uint64_t>(), img + dict_off, dsize, d->V.as<uint64_t>());
    }
    DEC_TRY(d, hipEventRecord(d->ev[2], s));
    if (nrows && nused) {
        hipLaunchKernelGGL(k_dec_tile_last, dim3(grid_for(ntiles * ubatch)), dim3(256), 0, s, rows, d->rstart.as<uint64_t>(), (uint64_t)nrows, nused, ubatch,
                           d->V.as<uint64_t>(), d->last.as<uint64_t>(), d->has.as<uint8_t>(), ntiles);
        const uint64_t n1 = nsuper * nused;
        hipLaunchKernelGGL(k_dec_carry1, dim3((uint32_t)((n1 + 255) / 256)), dim3(256), 0, s, d->last.as<uint64_t>(), d->has.as<uint8_t>(), ntiles, nused, nsuper,
                           d->slast.as<uint64_t>(), d->shas.as<uint8_t>());
        hipLaunchKernelGGL(k_dec_carry2, dim3((nused + 255) / 256), dim3(256), 0, s, d->slast.as<uint64_t>(), d->shas.as<uint8_t>(), nsuper, nused);
        hipLaunchKernelGGL(k_dec_carry3, dim3((uint32_t)((n1 + 255) / 256)), dim3(256), 0, s, d->last.as<uint64_t>(), d->has.as<uint8_t>(), ntiles, nused, nsuper,
                           d->slast.as<uint64_t>());
    }
    DEC_TRY(d, hipEventRecord(d->ev[3], s));

    // ---- text: lengths, offsets, bytes ----
    const uint32_t cbatch = (ncols + 63) / 64;
    DEC_TRY(d, d->rowlen.ensure(4ull * ((uint64_t)nrows + 2)));
    DEC_TRY(d, d->rowoff.ensure(8ull * ((uint64_t)nrows + 2)));
    DEC_TRY(d, d->pre.ensure(4ull * (uint64_t)nrows * cbatch + 64));
    DEC_TRY(d, d->scan_tmp.ensure(scan_tmp_bytes((uint64_t)nrows + 1)));
    DecText a{};
    a.rows = rows; a.rstart = d->rstart.as<uint64_t>(); a.nrows = nrows; a.ncols = ncols; a.nused = nused;
    a.type = d->d_type.as<uint8_t>(); a.used_of = d->d_used_of.as<int32_t>(); a.ubase = d->d_ubase.as<uint64_t>();
    a.V = d->V.as<uint64_t>(); a.carry = d->last.as<uint64_t>();
    a.dict = img + dict_off; a.dsize = dsize;
    a.rowlen = d->rowlen.as<uint32_t>(); a.pre = d->pre.as<uint32_t>(); a.rowoff = d->rowoff.as<uint64_t>(); a.out = nullptr;
    uint64_t text_bytes = 0;
    if (nrows && ncols) {
        hipLaunchKernelGGL((k_dec_text<false>), dim3(grid_for(ntiles)), dim3(256), 0, s, a, ntiles, cbatch);
        launch_scan_u32_to_u64(d->rowlen.as<uint32_t>(), d->rowoff.as<uint64_t>(), nrows, d->scan_tmp.p, nullptr, s);
        DEC_TRY(d, hipMemcpyAsync(&text_bytes, d->rowoff.as<uint64_t>() + nrows, 8, hipMemcpyDeviceToHost, s));
        DEC_TRY(d, hipStreamSynchronize(s));
    }
    DEC_TRY(d, hipEventRecord(d->ev[4], s));
    DEC_TRY(d, d->text.ensure(text_bytes + 64));
    if (text_bytes) {
        a.out = d->text.as<uint8_t>();
        const size_t shm = 4 * DEC_ROWBUF;
        if (cbatch <= 8 && !experiment("ZDW_DEC_DIRECT")) {
            // batches per wave: 10 GB x 200 columns, 1 / 2 / 4 per wave: 21.8 / 21.4 / 20.9 ms (the walk is not short of waves);
            // schemas of 5..8 batches take two groups of 4 (8 in one wave: 207 registers, 2 waves per SIMD)
            uint32_t nb = cbatch <= 1 ? 1u : cbatch <= 2 ? 2u : 4u;
            if (const char *e = test_hook("ZDW_DEC_GROUP")) { const int v = atoi(e); nb = v >= 8 ? 8u : v >= 4 ? 4u : v >= 2 ? 2u : 1u; }      // (A/B measurements)
            while (nb > 1 && nb / 2 >= cbatch) nb /= 2;
            const uint32_t ngroups = (cbatch + nb - 1) / nb;
            const uint32_t g = grid_for(ntiles * ngroups);
            if (nb == 1) hipLaunchKernelGGL((k_dec_rows<1>), dim3(g), dim3(256), shm, s, a, ntiles, cbatch, ngroups);
            else if (nb == 2) hipLaunchKernelGGL((k_dec_rows<2>), dim3(g), dim3(256), shm, s, a, ntiles, cbatch, ngroups);
            else if (nb == 4) hipLaunchKernelGGL((k_dec_rows<4>), dim3(g), dim3(256), shm, s, a, ntiles, cbatch, ngroups);
            else hipLaunchKernelGGL((k_dec_rows<8>), dim3(g), dim3(256), shm, s, a, ntiles, cbatch, ngroups);
        } else {
            hipLaunchKernelGGL((k_dec_text<true>), dim3(grid_for(ntiles)), dim3(256), 0, s, a, ntiles, cbatch);
        }
    }
    DEC_TRY(d, hipEventRecord(d->ev[5], s));
    DEC_TRY(d, hipStreamSynchronize(s));
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return dfail(d, DEC_DEVICE, "decode launch: %s", hipGetErrorString(le));
    for (int i = 0; i < 5; ++i) { float ms = 0; (void)hipEventElapsedTime(&ms, d->ev[i], d->ev[i + 1]); d->stage_ms[i] = ms; }
    d->text_bytes = text_bytes;
    blk->text_bytes = text_bytes;
    blk->zdw_bytes = (ro - d->pos) + rows_end;
    d->pos = ro + rows_end;
    if (blk->is_last_block) d->ended = true;
    else if (d->pos >= d->len) return dfail(d, DEC_CORRUPT, "the stream ends after a block that is not marked last");
    d->have_block = true;
    return DEC_OK;
}

int zdw_dec_text_device(zdw_dec *d, const void **d_text, uint64_t *text_len)
{
    if (!d || !d->have_block) return d ? dfail(d, DEC_DEVICE, "no block decoded") : DEC_DEVICE;
    if (d_text) *d_text = d->text.p;
    if (text_len) *text_len = d->text_bytes;
    return DEC_OK;
}

int zdw_dec_text_write(zdw_dec *d, zdw_write_fn write, void *user, uint64_t *text_len)
{
    if (!d || !d->have_block) return d ? dfail(d, DEC_DEVICE, "no block decoded") : DEC_DEVICE;
    if (!write) return dfail(d, DEC_DEVICE, "zdw_dec_text_write: write is NULL");
    (void)hipSetDevice(d->device);
    for (int i = 0; i < 2; ++i) {
        if (!d->piece[i]) DEC_TRY(d, hipHostMalloc(&d->piece[i], PIECE, hipHostMallocDefault));
        if (!d->piece_ev[i]) DEC_TRY(d, hipEventCreateWithFlags(&d->piece_ev[i], hipEventDisableTiming));
    }
    const uint64_t total = d->text_bytes;
    const uint8_t *src = d->text.as<uint8_t>();
    auto issue = [&](uint64_t k) -> hipError_t {
        const uint64_t off = k * PIECE, n = std::min<uint64_t>(PIECE, total - off);
        hipError_t e = hipMemcpyAsync(d->piece[k & 1], src + off, n, hipMemcpyDeviceToHost, d->s);
        if (e == hipSuccess) e = hipEventRecord(d->piece_ev[k & 1], d->s);
        return e;
    };
    const uint64_t npieces = (total + PIECE - 1) / PIECE;
    if (npieces) DEC_TRY(d, issue(0));
    for (uint64_t k = 0; k < npieces; ++k) {
        DEC_TRY(d, hipEventSynchronize(d->piece_ev[k & 1]));
        if (k + 1 < npieces) DEC_TRY(d, issue(k + 1));
        const uint64_t off = k * PIECE, n = std::min<uint64_t>(PIECE, total - off);
        if (write(user, static_cast<const uint8_t *>(d->piece[k & 1]), n) != 0) { (void)hipStreamSynchronize(d->s); return dfail(d, DEC_DEVICE, "the writer stopped the transfer"); }
    }
    if (text_len) *text_len = total;
    return DEC_OK;
}

int zdw_dec_stage_ms(const zdw_dec *d, float ms[5])
{
    if (!d || !ms) return DEC_DEVICE;
    for (int i = 0; i < 5; ++i) ms[i] = d->stage_ms[i];
    return DEC_OK;
}

}  // extern "C"
