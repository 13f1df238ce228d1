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

// The dynamic LDS of the launching kernel (pass 1 keeps, per wave, a ring of the last two input rounds there).
extern __shared__ __attribute__((aligned(16))) uint8_t zdw_lds[];
constexpr uint32_t BYTE_RING = 2048;     // two 1 KiB rounds; 8 mirror bytes follow so that 8-byte reads may straddle the wrap

// Where the bytes of one field are read from: the wave's LDS byte ring when the whole field is still in
// it (the common case: fields are consumed within a round or two of being tokenised), else global memory.
// LDS reads do not go through the vector L1, whose per-lane request rate bounds the field work.
struct FieldBytes {
    const uint8_t *g;        // field start in global memory
    const uint8_t *g_end;    // end of that allocation (reads never reach it)
    uint32_t ring;           // offset of the wave's byte ring in zdw_lds
    uint32_t at;             // (absolute field start) mod BYTE_RING
    bool in_lds;
    // bytes [i, i + want) of the field as a little-endian word, upper bytes zero (want in 1..8)
    __device__ __forceinline__ uint64_t word(uint32_t i, uint32_t want) const
    {
        if (in_lds) {
            const uint64_t w = reinterpret_cast<const PackedU64 *>(zdw_lds + ring + ((at + i) & (BYTE_RING - 1)))->v;
            return want >= 8 ? w : (w & ((1ull << (8 * want)) - 1ull));
        }
        return load_upto8(g + i, want, g_end);
    }
    __device__ __forceinline__ uint32_t byte(uint32_t i) const
    {
        return in_lds ? zdw_lds[ring + ((at + i) & (BYTE_RING - 1))] : g[i];
    }
};
__device__ __forceinline__ FieldBytes global_bytes(const uint8_t *p, const uint8_t *end)
{
    FieldBytes f;
    f.g = p; f.g_end = end; f.ring = 0; f.at = 0; f.in_lds = false;
    return f;
}

// Does the stored occurrence at image position `pos` (same first `len` bytes as the candidate) END
// after `len` bytes?  The next byte must be the field's separator; it is a separator iff the run of
// backslashes before it is even, and that run is the candidate's own trailing run (identical bytes),
// so its parity is passed in.  With -t the stored raw field may carry trailing blanks before its
// separator.  `next` is the byte at pos+len when the caller already holds it (else < 0).
__device__ inline bool stored_ends_here(const uint8_t *in, uint64_t pos, uint32_t len, uint64_t hi, bool trim, bool tail_bs_odd, int next)
{
    uint64_t q = pos + len;
    if (q >= hi) return false;
    uint32_t b = next >= 0 ? (uint32_t)next : in[q];
    bool blanks = false;
    if (trim) while (b == ' ') { ++q; blanks = true; if (q >= hi) return false; b = in[q]; }
    if (b != '\t' && b != '\n') return false;
    return blanks || !tail_bs_odd;
}

// A string's identity for the table: its first 24 bytes in registers (zero padded; strings hold no NUL, so
// for len <= 24 the three words ARE the string), a 64-bit hash of all of it, parity of its trailing backslashes.
struct StrKey {
    uint64_t c0, c1, c2, h;
    bool tail_bs_odd;
};

__device__ inline void make_key(const FieldBytes &f, uint32_t len, StrKey &k)
{
    k.c0 = f.word(0, len < 8 ? len : 8u);
    k.c1 = len > 8 ? f.word(8, len - 8 < 8 ? len - 8 : 8u) : 0ull;
    k.c2 = 0;
    uint64_t h = 0x9E3779B97F4A7C15ull ^ len;
    h = (h ^ k.c0) * 0xff51afd7ed558ccdull; h ^= h >> 32;
    uint64_t last = k.c0;
    if (len > 8) { h = (h ^ k.c1) * 0xff51afd7ed558ccdull; h ^= h >> 32; last = k.c1; }
    for (uint32_t i = 16; i < len; i += 8) {
        last = f.word(i, len - i < 8 ? len - i : 8u);
        if (i == 16) k.c2 = last;
        h = (h ^ last) * 0xff51afd7ed558ccdull;
        h ^= h >> 32;
    }
    k.h = mix64(h);
    k.tail_bs_odd = false;
    if (((last >> (8 * ((len - 1) & 7u))) & 0xffu) == '\\') {
        uint32_t nbs = 1;
        while (nbs < len && f.byte(len - 1 - nbs) == '\\') ++nbs;
        k.tail_bs_odd = nbs & 1u;
    }
}

// Insert-or-find in the block dictionary table.
//
// A slot is 32 bytes, 32-byte aligned (one L2 sector pair):
//     word 0   key = tag(24) | first-occurrence position + 1 (40); 0 = empty.  Claimed with a 64-bit CAS.
//     word 1-3 the string's first 24 bytes, zero padded, written by the claimer AFTER the CAS:
//              words 3, 2 first, then (every earlier store drained) word 1.  Strings hold no NUL and
//              are never empty, so word 1 != 0 means "words 1-3 are complete".
// A lookup reads the whole slot with two 16-byte L2-coherent loads issued in order (key + word 1 first).
// When word 1 is set, strings of <= 24 bytes are decided from the slot alone -- one random memory access
// per lookup instead of two (the second one, into the image at the first occurrence, was the largest
// single cost of pass 1).  While word 1 is still zero, or for longer strings, the bytes are compared
// against the first occurrence in the image (or the imported-string buffer): always exact.
// Returns slot index + 1, or 0 when the table is too full (caller retries with more slots).
// `n` = allocation size of the TSV image (reads never reach it); imported strings sit in a buffer with
// >= 16 bytes of slack.  The caller counts new entries per wave (is_new) and adds them to st->entries in
// bulk: one returning atomic per insert on a single counter serialises the whole chip (~70 M/s).
struct SlotWords { uint64_t key, s0, s1, s2; };

__device__ __forceinline__ SlotWords load_slot(const uint64_t *slot)
{
    // one asm block: the two loads issue in program order and return in order; sc1 = served by the L2
    // (coherent across CUs), so a set word 1 implies words 2-3 stored before it are seen as well
    uint4 lo, hi;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(slot) : "memory");
    SlotWords w;
    w.key = ((uint64_t)lo.y << 32) | lo.x; w.s0 = ((uint64_t)lo.w << 32) | lo.z;
    w.s1 = ((uint64_t)hi.y << 32) | hi.x; w.s2 = ((uint64_t)hi.w << 32) | hi.z;
    return w;
}

__device__ inline uint32_t dict_probe(const uint8_t *in, uint64_t n, const uint8_t *extra, uint64_t hi, uint64_t *slots,
                                      uint32_t *slot_len, uint64_t slot_mask, P1Status *st, uint64_t pos, uint32_t len, bool trim,
                                      const StrKey &key, const FieldBytes &f, bool &is_new)
{
    const uint64_t h = key.h, c0 = key.c0, c1 = key.c1, c2 = key.c2;
    const bool tail_bs_odd = key.tail_bs_odd;
    if (st->pad == 16u) return (uint32_t)(h & slot_mask) + 1;   // profiling only (ZDW_DEBUG_SKIP=16): no table access
    const bool trust_tag = st->pad == 32u;                        // profiling only (ZDW_DEBUG_SKIP=32): no string compare
    const uint64_t tag = h >> POS_BITS;
    const uint64_t mine = (tag << POS_BITS) | (pos + 1);
    uint64_t i = h & slot_mask;
    const uint32_t max_probe = slot_mask + 1 < 8192 ? (uint32_t)(slot_mask + 1) : 8192u;
    for (uint32_t probe = 0; probe < max_probe; ++probe) {
        uint64_t *slot = slots + 4 * i;
        SlotWords w = load_slot(slot);
        if (w.key == 0) {
            const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(slot), 0ull, (unsigned long long)mine);
            if (old == 0) {
                slot_len[i] = len;
                // publish the inline copy: words 3 and 2, drain, then word 1
                if (c2) __hip_atomic_store(slot + 3, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c1) __hip_atomic_store(slot + 2, c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(slot + 1, c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                is_new = true;
                return (uint32_t)i + 1;
            }
            w.key = old; w.s0 = 0;                        // somebody else just claimed it: its inline copy is not trusted yet
        }
        if ((w.key >> POS_BITS) == tag) {
            const uint64_t sp = (w.key & POS_MASK) - 1;
            if (sp == pos || trust_tag) return (uint32_t)i + 1;
            bool decided = false, same = false;
            if (w.s0 != 0) {
                // inline copy complete: it IS the stored string when that is <= 24 bytes long, else its first 24 bytes
                same = w.s0 == c0 && w.s1 == c1 && w.s2 == c2;
                decided = !same || len <= 24;
                // (same && len <= 24): equal padded words => equal strings, because neither holds a NUL;
                // a stored string longer than 24 bytes has a non-zero byte 24.. but then c2's padding would differ
                // only if len < 24 -- for len == 24 exactly the stored one may continue: check below
                if (same && len == 24) decided = false;
            }
            if (decided) { if (same) return (uint32_t)i + 1; }
            else if (sp >= EXTRA_BASE) {
                // imported string (NUL terminated, >= 16 bytes of slack behind it): a shorter one shows
                // its NUL inside the compared window, so the walk stops before leaving it
                const uint8_t *q = extra + (sp - EXTRA_BASE);
                bool eq = true;
                for (uint32_t k = 0; k < len && eq; k += 8) {
                    const uint32_t want = len - k < 8 ? len - k : 8u;
                    const uint64_t m = want >= 8 ? ~0ull : ((1ull << (8 * want)) - 1ull);
                    const uint64_t cw = k == 0 ? c0 : (k == 8 ? c1 : (k == 16 ? c2 : f.word(k, want)));
                    eq = (load8(q + k) & m) == cw;
                }
                if (eq && q[len] == 0) return (uint32_t)i + 1;
            } else if (sp + len < hi) {
                const uint8_t *q = in + sp;
                const uint8_t *q_end = in + n;
                bool eq = true;
                int next = -1;                              // the stored byte right after the compared bytes, when it came along
                for (uint32_t k = 0; k < len && eq; k += 8) {
                    const uint32_t want = len - k < 8 ? len - k : 8u;
                    const uint64_t cw = k == 0 ? c0 : (k == 8 ? c1 : (k == 16 ? c2 : f.word(k, want)));
                    if (q + k + 8 <= q_end) {
                        const uint64_t sw = load8(q + k);
                        if (want < 8) { eq = (sw & ((1ull << (8 * want)) - 1ull)) == cw; next = (int)((sw >> (8 * want)) & 0xffu); }
                        else eq = sw == cw;
                    } else {
                        eq = load_upto8(q + k, want, q_end) == cw;
                    }
                }
                if (eq && stored_ends_here(in, sp, len, hi, trim, tail_bs_odd, next)) return (uint32_t)i + 1;
            }
        }
        i = (i + 1) & slot_mask;
    }
    st->overflow = 1;
    return 0;
}

__device__ inline uint32_t dict_insert(const FieldBytes &f, const uint8_t *in, uint64_t n, const uint8_t *extra, uint64_t hi,
                                       uint64_t *slots, uint32_t *slot_len, uint64_t slot_mask, P1Status *st, uint64_t pos, uint32_t len,
                                       bool trim, bool &is_new)
{
    StrKey key;
    make_key(f, len, key);
    return dict_probe(in, n, extra, hi, slots, slot_len, slot_mask, st, pos, len, trim, key, f, is_new);
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
