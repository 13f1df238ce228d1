// zdw_dict_dev.h — device side of the block dictionary table (shared by pass 1 and the
// multi-GPU string import).  Replaces Dictionary::insert / StringHeap (cplusplus/dictionary.cpp:31-51,
// stringheap.cpp:31-86): an open-addressing table of first occurrences instead of a red-black tree
// of heap copies; strings stay where they are in the TSV image.
#pragma once
#include "zdw_dev.h"
#include "zdw_kernels.h"

namespace zdw {

__device__ __forceinline__ const uint8_t *string_at(const uint8_t *in, const uint8_t *extra, uint64_t pos)
{
    return pos >= EXTRA_BASE ? extra + (pos - EXTRA_BASE) : in + pos;
}

// ---- unaligned 8-byte window reads (gfx950 runs in unaligned access mode: one global_load_dwordx2) ----
struct __attribute__((packed)) PackedU64 { uint64_t v; };
__device__ __forceinline__ uint64_t load8(const uint8_t *p) { return reinterpret_cast<const PackedU64 *>(p)->v; }
// the next `want` (1..8) bytes at p as a little-endian word, upper bytes zero; never touches `end` or beyond
__device__ __forceinline__ uint64_t load_upto8(const uint8_t *p, uint32_t want, const uint8_t *end)
{
    if (p + 8 <= end) {
        const uint64_t w = load8(p);
        return want >= 8 ? w : (w & ((1ull << (8 * want)) - 1ull));
    }
    uint64_t w = 0;
    for (uint32_t k = 0; k < want; ++k) w |= (uint64_t)p[k] << (8 * k);
    return w;
}

// Does the stored occurrence at position `pos` (same first `len` bytes as the candidate) END after
// `len` bytes?  Imported strings are NUL terminated.  In the TSV image the next byte must be the
// field's separator; it is a separator iff the run of backslashes before it is even, and that run is
// the candidate's own trailing run (identical bytes), so its parity is passed in.  With -t the stored
// raw field may carry trailing blanks before its separator.
__device__ inline bool stored_ends_here(const uint8_t *in, const uint8_t *extra, uint64_t pos, uint32_t len, uint64_t hi,
                                        bool trim, bool tail_bs_odd)
{
    if (pos >= EXTRA_BASE) return extra[pos - EXTRA_BASE + len] == 0;
    uint64_t q = pos + len;
    bool blanks = false;
    if (trim) while (q < hi && in[q] == ' ') { ++q; blanks = true; }
    if (q >= hi) return false;
    const uint8_t b = in[q];
    if (b != '\t' && b != '\n') return false;
    return blanks || !tail_bs_odd;
}

// Insert-or-find in the block dictionary table.  slot word = tag(24) | position+1 (40); 0 = empty.
// Returns slot index + 1, or 0 when the table is too full (caller retries with more slots).
// `n` = allocation size of the TSV image (reads never reach it); imported strings sit in a buffer
// with >= 16 bytes of slack.
__device__ inline uint32_t dict_insert(const uint8_t *in, uint64_t n, const uint8_t *extra, uint64_t hi, uint64_t *slots,
                                       uint32_t *slot_len, uint64_t slot_mask, P1Status *st, uint64_t pos, uint32_t len, bool trim,
                                       bool &is_new)
{
    // The caller counts new entries per wave (is_new) and adds them to st->entries in bulk: one
    // returning atomic per insert on a single counter serialises the whole chip (~70 M/s).
    const uint8_t *p = string_at(in, extra, pos);
    const uint8_t *p_end = pos >= EXTRA_BASE ? p + len + 16 : in + n;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
    for (uint32_t i = 0; i < len; i += 8) {
        const uint64_t w = load_upto8(p + i, len - i < 8 ? len - i : 8u, p_end);
        h = (h ^ w) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    h = mix64(h);
    bool tail_bs_odd = false;
    if (p[len - 1] == '\\') {
        uint32_t nbs = 1;
        while (nbs < len && p[len - 1 - nbs] == '\\') ++nbs;
        tail_bs_odd = nbs & 1u;
    }
    const uint64_t tag = h >> POS_BITS;
    const uint64_t mine = (tag << POS_BITS) | (pos + 1);
    uint64_t i = h & slot_mask;
    const uint32_t max_probe = slot_mask + 1 < 8192 ? (uint32_t)(slot_mask + 1) : 8192u;
    for (uint32_t probe = 0; probe < max_probe; ++probe) {
        uint64_t w = __hip_atomic_load(&slots[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w == 0) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(&slots[i]), 0ull, (unsigned long long)mine);
            if (old == 0) {
                slot_len[i] = len;
                is_new = true;
                return (uint32_t)i + 1;
            }
            w = old;
        }
        if ((w >> POS_BITS) == tag) {
            const uint64_t sp = (w & POS_MASK) - 1;
            if (sp == pos) return (uint32_t)i + 1;
            const bool stored_extra = sp >= EXTRA_BASE;
            bool same = stored_extra || sp + len < hi;
            if (same) {
                // 8 bytes at a time.  An imported string shorter than `len` shows its NUL inside the
                // compared window (the candidate holds no NUL), so the walk stops before leaving it.
                const uint8_t *q = string_at(in, extra, sp);
                const uint8_t *q_end = stored_extra ? q + len + 16 : in + n;
                for (uint32_t k = 0; k < len; k += 8) {
                    const uint32_t want = len - k < 8 ? len - k : 8u;
                    if (load_upto8(q + k, want, stored_extra ? q + k + 8 : q_end) != load_upto8(p + k, want, p_end)) { same = false; break; }
                }
            }
            if (same && stored_ends_here(in, extra, sp, len, hi, trim, tail_bs_odd)) return (uint32_t)i + 1;
        }
        i = (i + 1) & slot_mask;
    }
    st->overflow = 1;
    return 0;
}

// big-endian 8-byte sort key of bytes [8*depth, 8*depth+8) of a string, zero padded.  Strings hold
// no NUL byte, so zero padding orders a shorter prefix first, like strcmp (dictionary.h:30-35).
__device__ inline uint64_t sort_key8(const uint8_t *p, uint32_t len, uint32_t depth)
{
    uint64_t k = 0;
    const uint64_t base = 8ull * depth;
#pragma unroll
    for (uint32_t b = 0; b < 8; ++b) {
        const uint64_t i = base + b;
        k = (k << 8) | (i < len ? p[i] : 0u);
    }
    return k;
}

}  // namespace zdw
