// zdw_dict_dev.h — device side of the block dictionary table (shared by pass 1 and the
// multi-GPU string import).  Replaces Dictionary::insert / StringHeap (cplusplus/dictionary.cpp:31-51,
// stringheap.cpp:31-86): an open-addressing table of first occurrences instead of a red-black tree
// of heap copies; strings stay where they are in the TSV image.
//
// A slot is 32 bytes, 32-byte aligned:
//     word 0   key = tag(16) << 40 | min(len, 255)(8) << 32 | unique id + 1 (32); 0 = empty.
//              Claimed with a 64-bit CAS.
//     word 1-3 the string's first 24 bytes, zero padded, written by the claimer AFTER the CAS:
//              words 3, 2 first, then (every earlier store drained) word 1.  Strings hold no NUL and
//              are never empty, so word 1 != 0 means "words 1-3 are complete".
// Unique ids are dense and handed out in order of first insertion (per-wave chunks of a global
// counter): uniq_pos[id] / uniq_len[id] hold where the string first occurs and how long it is; the
// claimer writes them before word 1, so a set word 1 also means "the unique's facts are there".  An id
// whose claim lost its race, or that was left over in a wave's chunk, stays a hole (length 0).  Early ids are
// the frequent strings, so pass 2's id -> dictionary offset gather hits a small, cache-resident
// prefix of its table.
// Strings of <= 24 bytes are decided from the slot alone (length byte + three words): one random
// memory access per lookup.  Longer strings, or a slot whose inline copy is not complete yet, are
// compared against the first occurrence in the image (or the imported-string buffer): always exact.
// The hash only places strings; equality never depends on it.
#pragma once
#include "zdw_dev.h"
#include "zdw_kernels.h"

namespace zdw {

constexpr uint32_t KEY_LEN_SHIFT = 32, KEY_TAG_SHIFT = 40;

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
// bytes [i, i + 8) of a string of `len` bytes at p, zero padded past its end
__device__ __forceinline__ uint64_t str_word(const uint8_t *p, uint32_t len, uint32_t i, const uint8_t *end)
{
    return i < len ? load_upto8(p + i, len - i < 8 ? len - i : 8u, end) : 0ull;
}

// The same without a branch (for loads that are to be in flight together): always one 8-byte load; when it would
// cross `end` it is moved back to end - 8 and shifted (the allocation must hold at least 8 bytes).
__device__ __forceinline__ uint64_t str_word_nb(const uint8_t *p, uint32_t len, uint32_t i, const uint8_t *end)
{
    const uint8_t *p8 = p + i;
    const uint8_t *q = p8 + 8 <= end ? p8 : end - 8;
    const uint64_t w = load8(q) >> ((8u * (uint32_t)(p8 - q)) & 63u);
    const uint32_t want = i < len ? (len - i < 8u ? len - i : 8u) : 0u;
    const uint64_t mask = want >= 8u ? ~0ull : ((1ull << (8u * want)) - 1ull);
    return w & mask;
}

// ---- placement hash: NH (UMAC) over the zero-padded string, 24 bytes per step -------------------
// (a + k)(b + k') on dword pairs costs one 32x32->64 multiply per EIGHT bytes; pass 1 evaluates the
// first (for most strings: only) step from registers.
constexpr uint32_t NH_K0 = 0x9E3779B9u, NH_K1 = 0x7F4A7C15u, NH_K2 = 0xF39CC060u, NH_K3 = 0x5CEDC834u, NH_K4 = 0x1082276Bu,
                   NH_K5 = 0xF86C6A11u, NH_FOLD = 0x85EBCA6Bu;
__device__ __forceinline__ uint64_t nh24(uint64_t c0, uint64_t c1, uint64_t c2)
{
    uint64_t acc = (uint64_t)((uint32_t)c0 + NH_K0) * (uint64_t)((uint32_t)(c0 >> 32) + NH_K1);
    acc += (uint64_t)((uint32_t)c1 + NH_K2) * (uint64_t)((uint32_t)(c1 >> 32) + NH_K3);
    acc += (uint64_t)((uint32_t)c2 + NH_K4) * (uint64_t)((uint32_t)(c2 >> 32) + NH_K5);
    return acc;
}
__device__ __forceinline__ uint32_t hash_place(uint64_t acc) { return (uint32_t)(acc >> 32) ^ __umulhi((uint32_t)acc, NH_FOLD); }
__device__ __forceinline__ uint32_t hash_tag(uint64_t acc) { return ((uint32_t)acc >> 16) & 0xffffu; }
__device__ __forceinline__ uint64_t make_slot_key(uint32_t tag16, uint32_t len, uint32_t id)
{
    return ((uint64_t)tag16 << KEY_TAG_SHIFT) | ((uint64_t)(len < 255u ? len : 255u) << KEY_LEN_SHIFT) | ((uint64_t)id + 1u);
}
// whole-string hash for any length (the first step equals what pass 1 computes from registers)
__device__ inline uint64_t nh_string(const uint8_t *p, uint32_t len, const uint8_t *end, uint64_t c0, uint64_t c1, uint64_t c2)
{
    uint64_t acc = nh24(c0, c1, c2);
    for (uint32_t i = 24; i < len; i += 24) {
        acc = (acc ^ (acc >> 31)) * 0xff51afd7ed558ccdull;
        acc += nh24(str_word(p, len, i, end), str_word(p, len, i + 8, end), str_word(p, len, i + 16, end));
    }
    return acc;
}

// A slot's 32 bytes: word 0 the key (tag, length byte, id + 1: claimed by compare-and-swap), word 1 the string's bytes 16..23 (s2),
// words 2-3 its bytes 0..15 (s0, s1: one 16-byte store).  Every word is written once.  The slot is COMPLETE when s0 is set and --
// for a string of more than 16 bytes, which the key's length byte tells -- s2 as well: the two stores need no order between them.
struct SlotWords { uint64_t key, s0, s1, s2; };
__device__ __forceinline__ bool slot_incomplete(const SlotWords &w)
{
    return w.s0 == 0 || ((((uint32_t)(w.key >> KEY_LEN_SHIFT)) & 0xffu) > 16u && w.s2 == 0);
}

// one asm block: the two loads issue in program order and return in order; sc1 = served at agent scope
// (coherent across CUs): the slot as it is
__device__ __forceinline__ SlotWords load_slot(const uint64_t *slot)
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 lo, hi;
    asm volatile("global_load_dwordx4 %0, %2, off sc1\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16 sc1\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(lo), "=&v"(hi) : "v"(slot) : "memory");
    SlotWords w;
    w.key = ((uint64_t)lo.y << 32) | lo.x; w.s2 = ((uint64_t)lo.w << 32) | lo.z;
    w.s0 = ((uint64_t)hi.y << 32) | hi.x; w.s1 = ((uint64_t)hi.w << 32) | hi.z;
    return w;
}

// The same in two steps: the loads are requested, other work runs, then the wave waits for them.  The compiler does not know that
// the registers fill in later: nothing may read `w` between the two calls (they are tied to the wait as in-out operands, so every
// use stays behind it), and the wait is for everything the wave has in flight.
struct SlotLoad { uint32_t lo __attribute__((ext_vector_type(4))); uint32_t hi __attribute__((ext_vector_type(4))); };
// These loads are ORDINARY ones (no sc1): they may be served from the CU's own vector L1, which other CUs' stores never refresh.
// A slot only ever goes empty -> claimed -> complete and then never changes (the table is cleared by another kernel, and a kernel
// starts with its L1 invalid), and every word of a slot is written once.  So a read that shows a COMPLETE slot (slot_incomplete)
// is right however old it is; anything else (empty, or claimed and not complete) says nothing and is read again at agent scope
// (load_slot) before it is acted on.  Measured (tools/probe_rate.hip): random 32-byte probes of an L2-resident table 252 G/s plain vs 130 G/s sc1, of an
// L1-resident one 440-540 vs 140-200, of a 1 GiB one 54 vs 46.
__device__ __forceinline__ void request_slot(const uint64_t *slot, SlotLoad &w)
{
    asm volatile("global_load_dwordx4 %0, %2, off\n\t"
                 "global_load_dwordx4 %1, %2, off offset:16"
                 : "=&v"(w.lo), "=&v"(w.hi) : "v"(slot) : "memory");
}
// The same request with the scope chosen per lane: the lanes of `agentm` (wave-uniform mask) read at agent scope -- what comes back
// is the slot as it is, an empty one included, and needs no second look -- the others as above.  For columns whose strings are mostly
// new (pass 1 keeps a flag per column): their probes find an empty slot, and an ordinary read of an empty slot only buys the second
// read.  Every lane's registers are written by exactly one of the two pairs of loads.
__device__ __forceinline__ void request_slot_scoped(const uint64_t *slot, SlotLoad &w, uint64_t agentm)
{
    uint64_t all;
    // (no lane of agentm -- tables whose strings repeat: the two ordinary loads and nothing else)
    asm volatile("s_cmp_eq_u64 %4, 0\n\t"
                 "s_cbranch_scc1 1f\n\t"
                 "s_mov_b64 %2, exec\n\t"
                 "s_andn2_b64 exec, %2, %4\n\t"
                 "global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n\t"
                 "s_and_b64 exec, %2, %4\n\t"
                 "global_load_dwordx4 %0, %3, off sc1\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16 sc1\n\t"
                 "s_mov_b64 exec, %2\n\t"
                 "s_branch 2f\n"
                 "1:\n\t"
                 "global_load_dwordx4 %0, %3, off\n\t"
                 "global_load_dwordx4 %1, %3, off offset:16\n"
                 "2:"
                 : "=&v"(w.lo), "=&v"(w.hi), "=&s"(all) : "v"(slot), "s"(agentm) : "memory", "scc");
}
__device__ __forceinline__ SlotWords await_slot(SlotLoad &l)
{
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(l.lo), "+v"(l.hi) : : "memory");
    SlotWords w;
    w.key = ((uint64_t)l.lo.y << 32) | l.lo.x; w.s2 = ((uint64_t)l.lo.w << 32) | l.lo.z;
    w.s0 = ((uint64_t)l.hi.y << 32) | l.hi.x; w.s1 = ((uint64_t)l.hi.w << 32) | l.hi.z;
    return w;
}

// ---- unique ids -------------------------------------------------------------------------------
struct IdAlloc { uint32_t cur, end, taken; };     // wave-uniform: ids [cur, end) belong to this wave; taken: all it has taken so far
#ifndef ZDW_ID_CHUNK
#define ZDW_ID_CHUNK 64
#endif
constexpr uint32_t ID_CHUNK = ZDW_ID_CHUNK;          // ids a pass-1 wave takes beyond what it needs right now

// One id for every active lane with `need` (call it with all lanes of the surrounding control flow).  Ids
// come from the wave's chunk; when that runs out the wave takes what is missing plus `extra` more from
// the global counter with one atomic -- more than `extra` once it has taken thousands: a sixteenth of what it
// has taken so far, up to 512 (the C2 shape, a new string per row: a million returning atomics on one word
// with chunks of 64, 15.8 -> 15.0 ms; the benchmark's waves take ~900 ids each and stay at 64 -- what is left
// of a chunk is holes among the ids, which the sort carries).
constexpr uint32_t ID_CHUNK_MAX = 512;
__device__ __forceinline__ uint32_t alloc_ids(IdAlloc &al, uint64_t m, P1Status *st, uint32_t extra);
__device__ __forceinline__ uint32_t alloc_ids(IdAlloc &al, bool need, P1Status *st, uint32_t extra) { return alloc_ids(al, (uint64_t)__ballot(need), st, extra); }
// ... for the lanes of mask m (wave-uniform)
__device__ __forceinline__ uint32_t alloc_ids(IdAlloc &al, uint64_t m, P1Status *st, uint32_t extra)
{
    if (!m) return 0;
    const uint32_t k = popc64(m), rank = mbcnt64(m), have = al.end - al.cur;
    uint32_t id = al.cur + rank;
    if (k <= have) {
        al.cur += k;
    } else {
        const int first = lowest_bit(m);
        const uint32_t more = al.taken >> 4;
        const uint32_t grab = k - have + (extra ? (more > extra ? (more < ID_CHUNK_MAX ? more : ID_CHUNK_MAX) : extra) : 0u);
        al.taken += grab;
        uint32_t fresh = 0;
        if (lane_id() == (uint32_t)first) fresh = atomicAdd(&st->id_next, grab);
        fresh = lane_value(fresh, first);
        if (rank >= have) id = fresh + (rank - have);
        al.cur = fresh + (k - have);
        al.end = fresh + grab;
    }
    return id;
}
// After a divergent region in which some lanes (mask m, non-empty) took ids: every lane adopts the chunk
// state of one that was inside.
__device__ __forceinline__ void adopt_ids(IdAlloc &al, uint64_t m)
{
    const int src = lowest_bit(m);
    al.cur = lane_value(al.cur, src);
    al.end = lane_value(al.end, src);
    al.taken = lane_value(al.taken, src);
}
// ids left in the wave's chunk stay holes
__device__ __forceinline__ void retire_ids(const IdAlloc &al, uint32_t *uniq_len, uint32_t id_cap)
{
    for (uint32_t id = al.cur + lane_id(); id < al.end; id += 64)
        if (id < id_cap) uniq_len[id] = 0;
}

// claim an empty slot for unique `id`: the unique's facts (in flight while the CAS travels: their lines are cold), the CAS, then the
// inline copy -- bytes 16..23 (if any) and bytes 0..15, in that order but without a wait between them (slot_incomplete looks at
// both).  Returns the previous key word (0 = claimed).  A lost race leaves the id a hole.
// FACTS_AT_AGENT_SCOPE (the general form: strings of more than 24 bytes are compared against their first occurrence while the pass
// runs, through uniq_pos / uniq_len): the facts must be seen by whoever sees the slot complete, so the wave waits for them (and for
// bytes 16..23) before the store that completes it.
// WIDE: bytes 0..15 go out as one 16-byte store.  Four consecutive registers: the lane-per-row kernel, which keeps its own loads in
// flight in registers the allocator must leave alone, has none to spare -- it stores bytes 8..15, waits, and completes with bytes 0..7.
template <bool FACTS_AT_AGENT_SCOPE, bool WIDE = true>
__device__ __forceinline__ uint64_t claim_slot(uint64_t *slot, uint64_t key, uint32_t id, uint64_t pos, uint32_t len, uint64_t *uniq_pos,
                                               uint32_t *uniq_len, uint64_t c0, uint64_t c1, uint64_t c2, bool track_first = false,
                                               uint64_t *uniq_key = nullptr)
{
    // (read after the kernel only, and only for ids that stay: ordinary stores)
    if (uniq_key) { uint64_t *k = uniq_key + 3ull * id; k[0] = c0; k[1] = c1; k[2] = c2; }
    // Only strings of more than 24 bytes are ever compared against their first occurrence while the pass
    // runs; for the others the facts are read after the kernel, so ordinary (write-back) stores do.
    if (track_first) {
        // (an earlier occurrence that another wave finds in the table a moment from now may get here first: a minimum, not a store)
        atomicMin(reinterpret_cast<unsigned long long *>(uniq_pos + id), (unsigned long long)pos);
        __hip_atomic_store(uniq_len + id, len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else if (FACTS_AT_AGENT_SCOPE) {
        __hip_atomic_store(uniq_pos + id, pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(uniq_len + id, len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        uniq_pos[id] = pos;
        uniq_len[id] = len;
    }
    const uint64_t old = atomicCAS(reinterpret_cast<unsigned long long *>(slot), 0ull, (unsigned long long)key);
    if (old == 0) {
        if (c2) __hip_atomic_store(slot + 1, c2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (WIDE) {
            if (FACTS_AT_AGENT_SCOPE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (sc1: agent scope, as the compiler emits for the store above)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 v = {(uint32_t)c0, (uint32_t)(c0 >> 32), (uint32_t)c1, (uint32_t)(c1 >> 32)};
            asm volatile("global_store_dwordx4 %0, %1, off offset:16 sc1" : : "v"(slot), "v"(v) : "memory");
        } else {
            if (c1) __hip_atomic_store(slot + 3, c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(slot + 2, c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else if (FACTS_AT_AGENT_SCOPE) {
        __hip_atomic_store(uniq_len + id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        uniq_len[id] = 0u;
    }
    return old;
}

// Does the stored occurrence at image position `pos` (same first `len` bytes as the candidate) END
// after `len` bytes?  Only needed when both lengths are >= 255 (the key's length byte saturates).  The
// next byte must be the field's separator; it is a separator iff the run of backslashes before it is
// even, and that run is the candidate's own trailing run (identical bytes), so its parity is passed in.
// With -t the stored raw field may carry trailing blanks before its separator.
__device__ inline bool stored_ends_here(const uint8_t *in, uint64_t pos, uint32_t len, uint64_t hi, bool trim, bool tail_bs_odd)
{
    uint64_t q = pos + len;
    if (q >= hi) return false;
    uint32_t b = in[q];
    bool blanks = false;
    if (trim) while (b == ' ') { ++q; blanks = true; if (q >= hi) return false; b = in[q]; }
    if (b != '\t' && b != '\n') return false;
    return blanks || !tail_bs_odd;
}

// Insert-or-find, general form: string of `len` >= 1 bytes at p (image or imported buffer; `end` = end of
// that allocation, reads never reach it), position code `pos`.  Returns unique id + 1, or 0 when the
// table is too full (caller retries with more slots).  Wave-level: every lane of the surrounding control
// flow calls it together (ids are handed out by ballot).  The caller counts new entries (is_new) and adds
// them to st->entries in bulk: one returning atomic per insert on a single counter serialises the chip.
struct DictTable {
    uint64_t *slots; uint64_t slot_mask;
    uint64_t *uniq_pos; uint32_t *uniq_len; uint32_t id_cap;
    uint64_t *uniq_key;        // optional (Pass1Args.uniq_key)
    P1Status *st;
    uint32_t track_first;      // uniq_pos[id] must end up as the string's FIRST occurrence in the image (the memory-limit model needs the
                               // order in which the reference meets the strings): every occurrence takes part in a minimum.  The
                               // array starts as all ones.
};

// an occurrence of unique id `id` (0-based) at image position pos: keep the smallest (ordinary read first: the value only falls)
__device__ __forceinline__ void note_occurrence(uint64_t *uniq_pos, uint32_t id, uint64_t pos)
{
    if (pos < uniq_pos[id]) atomicMin(reinterpret_cast<unsigned long long *>(uniq_pos + id), (unsigned long long)pos);
}

// BULK: the string is read 96 bytes at a time with all twelve loads of a round in flight together, hashed and
// compared from registers; the plain form reads 8 bytes at a time and waits for each (one memory latency per step:
// 72 dependent loads per 60-byte string in a kernel that does nothing else).  Costs ~40 VGPRs: for kernels that
// only do this.
constexpr uint32_t BULK_WORDS = 12;           // 96 bytes = four 24-byte hash steps per round
template <bool BULK = false>
__device__ inline uint32_t dict_insert(const uint8_t *p, const uint8_t *end, const uint8_t *in, uint64_t n, const uint8_t *extra, uint64_t hi,
                                       const DictTable &d, IdAlloc &al, uint32_t id_extra, uint64_t pos, uint32_t len, bool trim, bool &is_new)
{
    const bool bulk = BULK && end - p >= 8 && n >= 8;          // (the branch-free loads need 8 bytes of allocation)
    uint64_t c0, c1, c2, acc;
    if (BULK && bulk) {
        c0 = c1 = c2 = acc = 0;
        for (uint32_t base = 0; base < len; base += 8u * BULK_WORDS) {
            uint64_t cw[BULK_WORDS];
#pragma unroll
            for (uint32_t k = 0; k < BULK_WORDS; ++k) cw[k] = base + 8u * k < len ? str_word_nb(p, len, base + 8u * k, end) : 0ull;
            if (base == 0) { c0 = cw[0]; c1 = cw[1]; c2 = cw[2]; }
#pragma unroll
            for (uint32_t j = 0; j < BULK_WORDS / 3; ++j) {
                const uint32_t off = base + 24u * j;
                if (off == 0) acc = nh24(cw[0], cw[1], cw[2]);
                else if (off < len) {
                    acc = (acc ^ (acc >> 31)) * 0xff51afd7ed558ccdull;
                    acc += nh24(cw[3 * j], cw[3 * j + 1], cw[3 * j + 2]);
                }
            }
        }
    } else {
        c0 = str_word(p, len, 0, end); c1 = str_word(p, len, 8, end); c2 = str_word(p, len, 16, end);
        acc = nh_string(p, len, end, c0, c1, c2);
    }
    const uint32_t tag = hash_tag(acc), len8 = len < 255u ? len : 255u;
    if (len > 24u && len > d.st->max_len) atomicMax(&d.st->max_len, len);
    uint64_t i = hash_place(acc) & d.slot_mask;
    const uint32_t max_probe = d.slot_mask + 1 < 8192 ? (uint32_t)(d.slot_mask + 1) : 8192u;
    uint32_t result = 0;
    bool searching = true;
    // The loop is uniform over the calling lanes (it runs while any of them still searches) and ids are
    // handed out at loop level, so the wave's id chunk stays consistent across those lanes.
    for (uint32_t probe = 0; probe < max_probe; ++probe) {
        if (!__ballot(searching)) break;
        uint64_t *slot = d.slots + 4 * i;
        SlotWords w;
        w.key = 1; w.s0 = w.s1 = w.s2 = 0;
        if (searching) w = load_slot(slot);
        const uint32_t nid = alloc_ids(al, searching && w.key == 0, d.st, id_extra);
        if (!searching) continue;
        if (w.key == 0) {
            if (nid >= d.id_cap) { d.st->overflow = 1; searching = false; continue; }
            const uint64_t old = claim_slot<true>(slot, make_slot_key(tag, len, nid), nid, pos, len, d.uniq_pos, d.uniq_len, c0, c1, c2, d.track_first != 0, d.uniq_key);
            if (old == 0) { is_new = true; result = nid + 1; searching = false; continue; }
            w.key = old; w.s0 = 0;                        // somebody else just claimed it
        }
        if (((uint32_t)(w.key >> KEY_TAG_SHIFT) & 0xffffu) == tag && ((uint32_t)(w.key >> KEY_LEN_SHIFT) & 0xffu) == len8) {
            // a candidate whose claimer has not finished its copy: wait for it (the claimer waits for
            // nobody, and a claimer in this very wave has finished before its rivals get here)
            for (uint32_t spin = 0; slot_incomplete(w) && spin < (1u << 22); ++spin) w = load_slot(slot);
            if (slot_incomplete(w)) { d.st->overflow = 1; searching = false; continue; }      // cannot happen; never hang on it
            const uint32_t from = 24;                     // bytes [0, 24) are decided by the inline copy
            if (w.s0 == c0 && w.s1 == c1 && w.s2 == c2) {
                if (from >= len) {                                                             // len <= 24, equal length bytes: the slot decides
                    result = (uint32_t)w.key; searching = false;
                    if (d.track_first && pos < EXTRA_BASE) note_occurrence(d.uniq_pos, result - 1u, pos);
                    continue;
                }
                // compare the rest against the stored occurrence; with a length byte < 255 both lengths are
                // known to be equal, so only bytes are compared
                const uint64_t sp = __hip_atomic_load(d.uniq_pos + ((uint32_t)w.key - 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool is_extra = sp >= EXTRA_BASE;
                if (is_extra || sp + len < hi) {
                    const uint8_t *q = string_at(in, extra, sp);
                    const uint8_t *q_end = is_extra ? q + len + 16 : in + n;      // imported strings carry >= 16 bytes of slack
                    bool eq = true;
                    if (BULK && bulk) {
                        // bytes [24, len) of both strings, a round of up to 96 bytes of each in flight at a time
                        uint64_t diff = 0;
                        for (uint32_t base = 24; base < len && diff == 0; base += 8u * BULK_WORDS) {
#pragma unroll
                            for (uint32_t k = 0; k < BULK_WORDS; ++k)
                                diff |= base + 8u * k < len ? str_word_nb(q, len, base + 8u * k, q_end) ^ str_word_nb(p, len, base + 8u * k, end) : 0ull;
                        }
                        eq = diff == 0;
                    } else {
                        for (uint32_t k = from; k < len && eq; k += 8) {
                            const uint32_t want = len - k < 8 ? len - k : 8u;
                            eq = load_upto8(q + k, want, q_end) == load_upto8(p + k, want, end);
                        }
                    }
                    if (eq && len8 == 255u) {
                        if (is_extra) eq = q[len] == 0;
                        else {
                            uint32_t nbs = 0;
                            while (nbs < len && p[len - 1 - nbs] == '\\') ++nbs;
                            eq = stored_ends_here(in, sp, len, hi, trim, (nbs & 1u) != 0);
                        }
                    }
                    if (eq) {
                        result = (uint32_t)w.key; searching = false;
                        if (d.track_first && pos < EXTRA_BASE) note_occurrence(d.uniq_pos, result - 1u, pos);
                        continue;
                    }
                }
            }
        }
        i = (i + 1) & d.slot_mask;
    }
    if (searching) d.st->overflow = 1;
    return result;
}

// big-endian 8-byte sort key of bytes [8*depth, 8*depth+8) of a string, zero padded.  Strings hold
// no NUL byte, so zero padding orders a shorter prefix first, like strcmp (dictionary.h:30-35).
// `end` = end of the allocation the string lives in: one unaligned 8-byte load when that stays inside.
__device__ inline uint64_t sort_key8(const uint8_t *p, uint32_t len, uint32_t depth, const uint8_t *end)
{
    const uint32_t base = 8u * depth;
    if (base >= len) return 0ull;
    const uint32_t want = len - base < 8u ? len - base : 8u;
    const uint64_t w = load_upto8(p + base, want, end);          // first byte in the low bits
    return __builtin_bswap64(w);
}

}  // namespace zdw
