// k_pass1.hip — row scan and pass 1 of the ZDW block encode on gfx950: row discovery, tokenising,
// string dedupe into the block dictionary table, integer parse, per-column ranges.
//
// Replaces the per-row loop of ConvertToZDW::parseInput (cplusplus/ConvertToZDW.cpp:329-414)
// together with GetNextRow (getnextrow.cpp:26-84), GetDataRow/get_next_column
// (ConvertToZDW.cpp:265-323,1048-1067) and Dictionary::insert (dictionary.cpp:31-51).
//
// The kernel is bound by instruction issue and by the latency of one wave, not by HBM, so the design goal is
// few instructions per input byte at full lane occupancy, and no wait that other work could cover:
//   * the TSV image is cut into spans (16 KiB .. 1 MiB); a wave visit owns every row that STARTS in its span
//     and follows the last of them past the span end (no row-length limit);
//   * the wave streams 2 KiB rounds as two 1 KiB chunks: one coalesced 16-byte load per lane and chunk, kept
//     in a two-round LDS byte ring.  The tokeniser only finds the bytes that are a tab or a newline (one exact
//     SWAR test for both) -- WHICH of the two a boundary is, and whether a backslash escapes it (Q1, Q4), is
//     decided by the lane that later handles the field, from the byte ring (two byte reads; the exact parity of a
//     backslash run only where one precedes a boundary).  One wave prefix sum per ROUND (both chunks' counts
//     packed in one word) places every boundary in a per-wave LDS ring of 4-byte descriptors {position
//     relative to the visit};
//   * the loads are software-pipelined against the field work: chunk B of a round is requested before the
//     pending descriptors of the round before are handled, chunk A of the next round before this round's
//     boundaries are emitted and handled, so one chunk (4 VGPRs) is in flight at any time and neither is waited for;
//   * column and row numbers are NOT computed per byte: a field is the gap between two consecutive
//     descriptors, its column is its distance to the last line end before it and its row the number of line
//     ends before it -- two ballots per 64 fields.  Whenever 64 descriptors are pending the wave handles them
//     one per lane, so field work runs at full lane occupancy whatever the field-length distribution is.
//     Strings of <= 24 bytes and plain integers of <= 16 digits are served from the LDS byte ring in registers
//     (NH hash with three multiplies, one 32-byte table slot decides a lookup, 24-bit SWAR digit arithmetic);
//     everything else goes through the general form in zdw_field_dev.h / zdw_dict_dev.h;
//   * pass 2 never re-reads the text: each field leaves a fixed-width record (u32 unique id + 1
//     or u64 parsed value) in rec[row][column].
#include "zdw_dev.h"
#include "zdw_knobs.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"
#include "zdw_field_dev.h"

#include <cstddef>
#include <cstdlib>

namespace zdw {

// Ablation hooks of the profiling tools (work removed from the kernel: NOT a valid encode) exist only in a build made with
// -DZDW_EXPERIMENTS (tools/ build their own copy of the library); the shipped kernel has none of them.
#ifdef ZDW_EXPERIMENTS
#define P1_ABLATE(a, bit) (((a).debug_skip & (bit)) != 0u)
#else
#define P1_ABLATE(a, bit) false
#endif

// Arguments the kernel needs once per span visit or on rare paths are NOT kept in scalar registers for its whole run (the kernel holds
// ~170 wave-uniform values against 102 SGPRs; what does not fit is spilled into lanes of a vector register and read back with
// v_readlane + wait states -- and the allocator picked the hash table's pointers for that).  They are read from the kernel-argument
// segment where they are used: the pointer passes through an empty asm statement, so the compiler cannot hoist the load to the
// kernel's entry.  (The kernel's first parameter is the Pass1Args: offset 0 of the segment.)
template <typename T>
__device__ __forceinline__ T kernarg_at(uint32_t off)
{
    typedef __attribute__((address_space(4))) const uint8_t *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<__attribute__((address_space(4))) const T *>(ka + off);
}
#define COLD(field) kernarg_at<decltype(Pass1Args::field)>((uint32_t)offsetof(Pass1Args, field))

constexpr uint32_t WROUND = 1024;                 // bytes per load step (a chunk): one 16-byte load per lane
constexpr uint32_t ROUND = 2 * WROUND;            // bytes per tokeniser round: chunks A and B
constexpr uint32_t DQ_CHUNK = 256;                // deferred-queue entries a wave takes per atomic
constexpr uint32_t BYTE_RING = 4096;              // two rounds of input bytes per wave
constexpr uint32_t RING_TAIL = 32;                // copy of the byte ring's first bytes behind its end: 24-byte reads never wrap
constexpr uint32_t RING_HEAD = 16;                // copy of its last bytes in front of it: the byte before position 0 of the ring
constexpr uint32_t RING_STRIDE = RING_HEAD + BYTE_RING + RING_TAIL;
#ifndef P1_MIN_WAVES
#define P1_MIN_WAVES 6          // waves per SIMD the register allocation of the main kernel aims at
#endif
constexpr uint32_t P1_WPB = 8;                    // independent waves per workgroup (they share the column tables): the default; launch_pass1 takes the
                                                  // number that leaves the most waves resident for the schema's tables (blockDim.x / 64 in the kernel)
constexpr uint32_t P1_WPB_MAX = 16;
constexpr uint32_t LUT_ROWS = 25;                 // byte masks "first L bytes" for L = 0..24, 32 bytes per row
// Descriptor ring entries per wave (u32 each): 512 for ordinary tables (a 2 KiB round holds ~220 boundaries); 1024 for
// tables of mostly empty fields (a round of such data holds 1000+), whose boundaries are then emitted chunk by chunk
// (~540 each).  What still does not fit is emitted in 16-lane pieces (<= 256 boundaries each), the ring drained in between.
constexpr uint32_t QCAP_SMALL = 512, QCAP_LARGE = 1024;
// A descriptor is the boundary's position relative to the visit.  Kept as its low 16 bits where nothing else rides on it: every pending
// boundary lies less than 64 KiB below the end of what the tokeniser has emitted (WaveState.pos_hi; a round is only emitted into a ring
// that still holds something when the boundary before the ring's head is that close -- otherwise the ring is drained first), so the
// consumer rebuilds the position from the low bits.  Half the ring: 5 KiB of LDS per wave instead of 6, which is what decides how
// many waves a CU holds.  (Tables of mostly empty fields: 24 bits of position and 8 of ordinal, four bytes.)
__host__ __device__ constexpr uint32_t desc_bytes(bool skip) { return skip ? 4u : 2u; }
constexpr uint32_t DESC_POS = 0xffffffu;
// descriptors that may stay pending behind a chunk A / behind a round (the fields of what stays are handled a step later, while the
// next chunk travels)
#ifndef P1_KEEP_AFTER_A
#define P1_KEEP_AFTER_A 63u
#endif
#ifndef P1_KEEP_AFTER_B
#define P1_KEEP_AFTER_B 127u
#endif          // a descriptor's position bits (relative to the visit; the visit moves its base before they run out)

// ------------------------------------------------------------------------------------------
// per-lane byte classes of 16 bytes
// ------------------------------------------------------------------------------------------
// 4-bit set of the bytes of w equal to the byte replicated in `pat`
__device__ __forceinline__ uint32_t eq4(uint32_t w, uint32_t pat)
{
    const uint32_t x = w ^ pat;                                        // zero byte where equal
    const uint32_t t = ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x;         // bit 7 set iff the byte is non-zero (exact)
    const uint32_t m = ~t & 0x80808080u;
    return ((m >> 7) * 0x00204081u >> 21) & 0xfu;                      // bits 7,15,23,31 -> 0..3
}
__device__ __forceinline__ uint32_t eq16(const uint4 &v, uint32_t pat)
{
    return eq4(v.x, pat) | (eq4(v.y, pat) << 4) | (eq4(v.z, pat) << 8) | (eq4(v.w, pat) << 12);
}

// bit 7 of each byte: set iff the byte of x is non-zero (exact, no carries between bytes)
__device__ __forceinline__ uint32_t nz7b(uint32_t x) { return ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x; }

// this lane's 16 bytes of the round at `pos`, branch-free: lanes whose chunk would start at or past
// the end of the allocation re-read the last 16-byte chunk that starts inside it (an aligned chunk
// whose first byte is inside an allocation never leaves its page); callers mask by position.
__device__ __forceinline__ uint4 load_round16(const uint8_t *in, uint64_t pos, uint32_t lane, uint64_t n)
{
    const uint64_t last = (n - 1) & ~15ull;                    // n > 0
    const uint64_t base = pos <= last ? pos : last;            // wave-uniform
    const uint64_t room = last - base;
    const uint32_t maxoff = room < WROUND - 16 ? (uint32_t)room : WROUND - 16;
    const uint32_t off = 16u * lane < maxoff ? 16u * lane : maxoff;
#ifdef ZDW_P1_COUNT_PLAIN_LOADS
    return *reinterpret_cast<const uint4 *>(in + base + off);
#else
    // (read once per pass: a non-temporal load -- the count's stream does not push the table's hot slots and the records' lines out of the caches)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + base + off));
    return make_uint4(x.x, x.y, x.z, x.w);
#endif
}

// bits i of a lane whose absolute position lp + i lies in [vlo, vhi)
__device__ __forceinline__ uint32_t valid16(uint64_t lp, uint64_t vlo, uint64_t vhi)
{
    const uint32_t lo_i = vlo > lp ? (vlo - lp < 16 ? (uint32_t)(vlo - lp) : 16u) : 0u;
    const uint32_t hi_i = vhi > lp ? (vhi - lp < 16 ? (uint32_t)(vhi - lp) : 16u) : 0u;
    return hi_i > lo_i ? (((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u)) : 0u;
}

// the same on 32-bit positions relative to a visit base
__device__ __forceinline__ uint32_t valid16r(uint32_t lp, uint32_t vlo, uint32_t vhi)
{
    const uint32_t lo_i = vlo > lp ? (vlo - lp < 16u ? vlo - lp : 16u) : 0u;
    const uint32_t hi_i = vhi > lp ? (vhi - lp < 16u ? vhi - lp : 16u) : 0u;
    return hi_i > lo_i ? (((1u << hi_i) - 1u) & ~((1u << lo_i) - 1u)) : 0u;
}

struct Lane16 {
    uint32_t tab, nl;     // UNESCAPED tabs / newlines among this lane's 16 bytes
};

// classification + escape resolution (special iff an even number of backslashes
// precedes).  bs_carry: the byte before this round is a backslash (wave-uniform, in/out).
__device__ __forceinline__ void classify16(const uint8_t *in, uint64_t lo, uint64_t n, uint64_t lp, const uint4 &v, uint32_t lane,
                                           uint32_t &bs_carry, Lane16 &m, bool want_tabs)
{
    const uint32_t bs = eq16(v, 0x5c5c5c5cu);
    m.nl = eq16(v, 0x0a0a0a0au);
    m.tab = want_tabs ? eq16(v, 0x09090909u) : 0u;
    uint32_t prev = wave_shr1(bs >> 15);
    if (lane == 0) prev = bs_carry;
    bs_carry = lane_value(bs >> 15, 63);
    const uint32_t cand = (m.tab | m.nl) & ((bs << 1) | prev) & 0xffffu;
    if (__ballot(cand != 0)) {          // rare: count the run exactly (bounded by the block start)
        uint32_t c = cand, esc = 0;
        while (c) {
            const uint32_t i = (uint32_t)__builtin_ctz(c);
            c &= c - 1;
            // (lanes past the end of the image hold re-read bytes: never follow those into memory)
            if (lp + i > lo && lp + i < n && (backslashes_before(in, lo, lp + i) & 1u)) esc |= 1u << i;
        }
        m.tab &= ~esc;
        m.nl &= ~esc;
    }
}

// ------------------------------------------------------------------------------------------
// effective end of the image = last row terminator + 1 (an unterminated tail is dropped,
// getnextrow.cpp:67-69).  One wave walks backwards; normally the first round finds it.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_find_effective_end(const uint8_t *in, uint64_t n, uint64_t lo, uint64_t *out)
{
    const uint32_t lane = lane_id();
    uint64_t result = lo;
    if (n > lo) {
        uint64_t pos = (n - 1) & ~(uint64_t)(WROUND - 1);
        for (;;) {
            const uint64_t lp = pos + 16ull * lane;
            const uint4 v = load_round16(in, pos, lane, n);
            uint32_t bs_carry = (pos > lo && in[pos - 1] == '\\') ? 1u : 0u;
            Lane16 m;
            classify16(in, lo, n, lp, v, lane, bs_carry, m, false);
            const uint32_t nl = m.nl & valid16(lp, lo, n);
            const uint64_t b = __ballot(nl != 0);
            if (b) {
                const int hl = highest_bit(b);
                const uint32_t hb = lane_value(31u - (uint32_t)__builtin_clz(nl | 1u), hl);
                result = pos + 16ull * (uint32_t)hl + hb + 1;
                break;
            }
            if (pos <= lo) break;          // pos is a multiple of 1024, so pos > lo implies pos >= 1024
            pos -= WROUND;
        }
    }
    if (lane == 0) *out = result;
}

void launch_find_effective_end(const uint8_t *in, uint64_t n, uint64_t lo, uint64_t *d_out, hipStream_t s)
{
    hipLaunchKernelGGL(k_find_effective_end, dim3(1), dim3(64), 0, s, in, n, lo, d_out);
}

// ------------------------------------------------------------------------------------------
// rows that start in each span.  A row start is a non-terminator byte that follows a row
// terminator (or the block start): blank lines are no rows (getnextrow.cpp:38-43).
// Streaming kernel: 1 byte read per input byte; every lane counts the starts among its own bytes
// and the wave adds up once per span.
// ------------------------------------------------------------------------------------------
// rows that start in [ss, se) (this wave's share of the count: the caller adds the lanes up); nul: some byte of it is NUL
__device__ __forceinline__ uint32_t count_row_starts(const uint8_t *__restrict__ in, uint64_t n, uint64_t lo, uint64_t ss, uint64_t se, uint32_t lane,
                                                     uint32_t &nul_out)
{
    uint32_t mine = 0, nul = 0;
    if (ss < se) {
        uint32_t prev_term = ((ss == lo) || is_row_terminator(in, lo, ss - 1)) ? 1u : 0u;
        const uint64_t pos0 = ss & ~(uint64_t)(WROUND - 1);
        uint32_t bs_carry = (pos0 > lo && in[pos0 - 1] == '\\') ? 1u : 0u;
        // four rounds in flight, the next four requested before these are counted: a wave that counts inside pass 1 has only
        // its own loads to wait for
        auto round = [&](const uint64_t pos, const uint4 v) {
            const uint64_t lp = pos + 16ull * lane;
            if (pos >= ss && pos + WROUND <= se) {
                // common case -- the round lies inside the span and no newline of it follows a backslash (so every
                // newline is a row terminator): count in place on the bit-7 flags of the SWAR compares (no compaction
                // to 16-bit sets)
                const uint32_t H = 0x80808080u, L1 = 0x01010101u;
                const uint32_t n0 = ~nz7b(v.x ^ 0x0a0a0a0au) & H, n1 = ~nz7b(v.y ^ 0x0a0a0a0au) & H, n2 = ~nz7b(v.z ^ 0x0a0a0a0au) & H,
                               n3 = ~nz7b(v.w ^ 0x0a0a0a0au) & H;                           // bit 7 of a byte: it is a newline
                const uint32_t b0 = v.x ^ 0x5c5c5c5cu, b1 = v.y ^ 0x5c5c5c5cu, b2 = v.z ^ 0x5c5c5c5cu, b3 = v.w ^ 0x5c5c5c5cu;
                const uint32_t anybs = (((b0 - L1) & ~b0) | ((b1 - L1) & ~b1) | ((b2 - L1) & ~b2) | ((b3 - L1) & ~b3)) & H;     // exact for "some byte is a backslash"
                bool plain = true;
                uint32_t carry_next = 0;
                if (__ballot(anybs != 0) | bs_carry) {
                    // backslashes about: is one of them right before a newline?
                    const uint32_t s0 = ~nz7b(b0) & H, s1 = ~nz7b(b1) & H, s2 = ~nz7b(b2) & H, s3 = ~nz7b(b3) & H;   // bit 7: the byte is a backslash
                    uint32_t sbefore = wave_shr1(s3);
                    if (lane == 0) sbefore = bs_carry ? 0x80000000u : 0u;
                    const uint32_t hit = (n0 & __builtin_amdgcn_alignbit(s0, sbefore, 24)) | (n1 & __builtin_amdgcn_alignbit(s1, s0, 24)) |
                                         (n2 & __builtin_amdgcn_alignbit(s2, s1, 24)) | (n3 & __builtin_amdgcn_alignbit(s3, s2, 24));
                    plain = __ballot(hit != 0) == 0;
                    carry_next = lane_value(s3 >> 31, 63);
                }
                if (plain) {
                    uint32_t before = wave_shr1(n3);                                          // the byte before my first one
                    if (lane == 0) before = prev_term ? 0x80000000u : 0u;
                    // "the previous byte is a newline" for each of my bytes, then: a row starts where that holds and the byte is none
                    mine += (uint32_t)__builtin_popcount(__builtin_amdgcn_alignbit(n0, before, 24) & ~n0)
                          + (uint32_t)__builtin_popcount(__builtin_amdgcn_alignbit(n1, n0, 24) & ~n1)
                          + (uint32_t)__builtin_popcount(__builtin_amdgcn_alignbit(n2, n1, 24) & ~n2)
                          + (uint32_t)__builtin_popcount(__builtin_amdgcn_alignbit(n3, n2, 24) & ~n3);
                    nul |= (((v.x - L1) & ~v.x) | ((v.y - L1) & ~v.y) | ((v.z - L1) & ~v.z) | ((v.w - L1) & ~v.w)) & H;        // exact for "some byte is NUL"
                    prev_term = lane_value(n3 >> 31, 63);
                    bs_carry = carry_next;
                    return;
                }
            }
            Lane16 m;
            classify16(in, lo, n, lp, v, lane, bs_carry, m, false);
            uint32_t vm = 0xffffu;
            if (pos < ss || pos + WROUND > se) vm = valid16(lp, ss, se);      // first / last round of the span
            const uint32_t nl = m.nl & vm;
            uint32_t pt = wave_shr1(nl >> 15) | ((nl << 1) & 0xffffu);
            // the byte before the first position of the span (or before this round) terminated a row
            if (prev_term) {
                const uint64_t first = pos > ss ? pos : ss;
                if (first >= lp && first < lp + 16) pt |= 1u << (uint32_t)(first - lp);
            }
            mine += (uint32_t)__builtin_popcount(pt & ~nl & vm);
            nul |= eq16(v, 0u) & vm;
            prev_term = lane_value(nl >> 15, 63);
        };
#ifndef P1_COUNT_AHEAD
#define P1_COUNT_AHEAD 4          // rounds of the count in flight (and as many more requested before these are counted)
#endif
        uint4 q[P1_COUNT_AHEAD];
#pragma unroll
        for (uint32_t k = 0; k < P1_COUNT_AHEAD; ++k) q[k] = load_round16(in, pos0 + k * WROUND, lane, n);
        for (uint64_t pos = pos0; pos < se; pos += P1_COUNT_AHEAD * WROUND) {
            uint4 c[P1_COUNT_AHEAD];
#pragma unroll
            for (uint32_t k = 0; k < P1_COUNT_AHEAD; ++k) c[k] = q[k];
#pragma unroll
            for (uint32_t k = 0; k < P1_COUNT_AHEAD; ++k) q[k] = load_round16(in, pos + (P1_COUNT_AHEAD + k) * WROUND, lane, n);
#pragma unroll
            for (uint32_t k = 0; k < P1_COUNT_AHEAD; ++k)
                if (k == 0 || pos + k * WROUND < se) round(pos + k * WROUND, c[k]);
        }
    }
    nul_out = nul;
    return mine;
}

__global__ __launch_bounds__(64) void k_count_row_starts(const uint8_t *__restrict__ in, uint64_t n, uint64_t lo, uint64_t hi,
                                                          uint64_t span0, uint32_t nspans, uint32_t span_shift,
                                                          uint32_t *__restrict__ span_starts, P1Status *st)
{
    const uint64_t SPAN_BYTES = 1ull << span_shift;
    const uint32_t lane = lane_id();
    for (uint32_t si = blockIdx.x; si < nspans; si += gridDim.x) {
        const uint64_t s0 = (span0 + si) << span_shift;
        const uint64_t ss = s0 > lo ? s0 : lo;
        const uint64_t se = s0 + SPAN_BYTES < hi ? s0 + SPAN_BYTES : hi;
        uint32_t nul = 0;
        const uint32_t mine = count_row_starts(in, n, lo, ss, se, lane, nul);
        uint32_t total;
        wave_excl_sum(mine, lane, total);
        const uint64_t anynul = __ballot(nul != 0);
        if (lane == 0) {
            span_starts[si] = total;
            if (anynul) st->nul_seen = 1;
        }
    }
}

void launch_count_row_starts(const uint8_t *in, uint64_t n, uint64_t lo, uint64_t hi, uint64_t span0, uint32_t nspans, uint32_t span_shift,
                             uint32_t *span_starts, P1Status *st, hipStream_t s)
{
    if (!nspans) return;
    uint32_t cap = 256u * 24u;
    if (const char *e = experiment("ZDW_SCAN_GRID")) cap = (uint32_t)atoi(e);      // tuning experiments
    const uint32_t grid = nspans < cap ? nspans : cap;
    hipLaunchKernelGGL(k_count_row_starts, dim3(grid), dim3(64), 0, s, in, n, lo, hi, span0, nspans, span_shift, span_starts, st);
}

// ------------------------------------------------------------------------------------------
// pass 1
// ------------------------------------------------------------------------------------------
// LDS of one workgroup (byte offsets into zdw_lds):
//   desc   P1_WPB x QCAP x 2 or 4   descriptor rings (size-aligned: ring addressing is one and-or); 16-bit descriptors except in the
//          form for tables of mostly empty fields (desc_bytes)
//   bytes  P1_WPB x (16 + 4 KiB + 32)   byte rings, each between a copy of its last 16 and of its first 32 bytes
//   lut    25 x 32 B        byte masks "first L bytes are 0xff" for L = 0..24
//   meta / mn / mx / seen   column tables shared by the waves of the workgroup (cols_in_lds)
struct P1Lds { uint32_t desc, bytes, lut, dqs, meta, mn, mx, seen, total; };

__host__ __device__ inline P1Lds p1_layout(uint32_t ncols, uint32_t nnum, bool cols_in_lds, uint32_t ring_bytes /* of one descriptor ring */, uint32_t wpb = P1_WPB)
{
    P1Lds L;
    L.desc = 0;
    L.bytes = wpb * ring_bytes;
    L.lut = L.bytes + wpb * RING_STRIDE;
    L.dqs = L.lut + LUT_ROWS * 32u;              // P1_WPB x {cur, end}: kept here, not in registers (the main loop has none to spare)
    L.meta = L.dqs + wpb * 8u;
    uint32_t end = L.meta;
    L.mn = L.mx = L.seen = end;
    if (cols_in_lds) {
        L.mn = (L.meta + 4u * ncols + 7u) & ~7u;          // 64-bit LDS atomics need 8-byte alignment
        L.mx = L.mn + 8u * nnum;
        L.seen = L.mx + 8u * nnum;
        end = L.seen + ncols;
    }
    L.total = (end + 15u) & ~15u;
    return L;
}

size_t pass1_lds_bytes(uint32_t ncols, uint32_t nnum, bool use_lds) { return p1_layout(ncols, nnum, use_lds, QCAP_SMALL * 4u).total; }

__device__ __forceinline__ uint32_t lds_u32(uint32_t off) { return *reinterpret_cast<const uint32_t *>(zdw_lds + off); }
__device__ __forceinline__ uint32_t lds_u16(uint32_t off) { return *reinterpret_cast<const uint16_t *>(zdw_lds + off); }
// the value in [ref - 65536, ref) whose low 16 bits are d16
__device__ __forceinline__ uint32_t near_below(uint32_t ref, uint32_t d16) { return ref - 1u - ((ref - 1u - d16) & 0xffffu); }
__device__ __forceinline__ uint64_t lds_u64(uint32_t off) { return *reinterpret_cast<const uint64_t *>(zdw_lds + off); }     // 8-byte aligned offsets only
// Field bytes sit at any byte offset of the ring.  A ds_read_b64 / _b128 off its natural alignment is replayed at ~64
// cycles per wave-instruction (SQ_LDS_UNALIGNED_STALL: 59 % of this kernel's LDS cycles before), so bytes are read
// as aligned dwords (ds_read2_b32 needs 4-byte alignment only) and moved into place with v_alignbyte.
template <uint32_t NW>      // NW dwords starting at byte offset `off` (any alignment): reads NW + 1 aligned dwords
__device__ __forceinline__ void lds_bytes(uint32_t off, uint32_t (&w)[NW])
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(zdw_lds + (off & ~3u));
    const uint32_t sh = off & 3u;
    uint32_t d[NW + 1];
#pragma unroll
    for (uint32_t k = 0; k <= NW; ++k) d[k] = q[k];
#pragma unroll
    for (uint32_t k = 0; k < NW; ++k) w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
}
__device__ __forceinline__ uint64_t u64_of(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }

// Flags in bit 7 of the bytes of four dwords (0x80 or 0) -> 16 bits in position order.  Two byte-wise dot products per pair of
// dwords (v_dot4_u32_u8: weights 1, 2, 4, 8 on the first dword, 16 .. 128 on the second, the first sum as the addend) give the
// pair's eight flags times 128: seven vector instructions for sixteen flags, where shifts, masks and a 24-bit multiply per dword
// took about thirty (this kernel is bound by the vector ALU's issue rate: DESIGN.md 4.1).
__device__ __forceinline__ uint32_t pack8x128(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_udot4(b, 0x80402010u, __builtin_amdgcn_udot4(a, 0x08040201u, 0u, false), false);
}
__device__ __forceinline__ uint32_t pack16(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return (pack8x128(a, b) >> 7) | (pack8x128(c, d) << 1);
}
// bit 7 of each byte of w: the byte is a tab (0x09) or a newline (0x0a).  y = byte ^ 8 maps them to 1 and 2; of the
// low seven bits m, m + 0x7f carries into bit 7 iff m >= 1 and m + 0x7d iff m >= 3 (no carry leaves a byte); bytes >= 0x80 are out.
__device__ __forceinline__ uint32_t tab_or_nl7(uint32_t w)
{
    const uint32_t y = w ^ 0x08080808u, m = y & 0x7f7f7f7fu;
    return (m + 0x7f7f7f7fu) & ~(m + 0x7d7d7d7du) & ~y & 0x80808080u;
}
// the 16-bit set, in position order, of the bytes of a lane's chunk that are a tab or a newline
__device__ __forceinline__ uint32_t boundaries16(const uint4 &v) { return pack16(tab_or_nl7(v.x), tab_or_nl7(v.y), tab_or_nl7(v.z), tab_or_nl7(v.w)); }

// wave-uniform state of one worker wave
// (measured: as 32-bit 0 / 1 values instead of bools these six cost pass 1 0.6 ms of 11 -- six more scalar registers in a kernel that is short of them)
typedef bool P1_FLAG;
struct WaveState {
    uint32_t desc, bytes;        // LDS offsets of this wave's rings (bytes: position 0 of the byte ring)
    uint32_t q_head, q_count;    // pending descriptors: ring indices q_head .. q_head + q_count - 1 (mod QCAP)
    uint32_t row_cur, col_cur;   // row / column of the field that ends at the head descriptor
    uint32_t ring_lo;            // smallest relative position the byte ring still holds
    uint32_t rel_se1;            // relative position of the span's last byte: the visit ends with the first line end at or past it
    uint32_t last_e;             // relative position of the last boundary that counted before the head descriptor (its field starts behind it)
    uint32_t last_ord;           // ... and its ordinal among the visit's boundaries (mod 256)
    uint32_t tok_ord;            // tokeniser: boundaries of the visit so far (descriptors carry it mod 256)
    uint32_t tok_prevb;          // tokeniser: the byte before the next chunk is a boundary
    P1_FLAG tok_bs;                 // tokeniser: the chunk before held a backslash
    P1_FLAG far;                    // the boundary at last_e lies further back than a descriptor can say: far_abs is its position
    uint64_t far_abs;
    P1_FLAG first_pending;          // the head descriptor is the visit's sentinel: the line end before the first owned row
    P1_FLAG ignoring;               // the visit began inside a row of the span before: everything up to that row's end is not ours
    P1_FLAG prev_nl;                // the boundary before the head descriptor is a line end
    P1_FLAG done;                   // the visit's last row has been handled
    uint64_t vbase;              // visit base: absolute position = vbase + relative position (multiple of BYTE_RING)
    uint64_t row_start;          // absolute position where the row that is open at the head descriptor began (m_LongestLine candidates)
    IdAlloc ids;                 // this wave's chunk of unique ids
    uint32_t rec_stride;         // Pass1Args.rec_stride, kept in a register (see k_pass1)
    uint32_t pos_hi;             // 16-bit descriptors: every pending boundary lies in [pos_hi - 65536, pos_hi) (desc_bytes)
};

// Handles `cnt` (1..64) descriptors from the ring head, one per lane.  Returns how many new dictionary
// entries were created.  DEFER: fields that need the general form are queued for k_pass1_deferred instead of
// being handled here (the general form costs this kernel a fifth of its registers and a wave per SIMD).
// TRIM: convertDWfile -t -- trailing blanks are cut off every field before anything else looks at it (ConvertToZDW.cpp:295-313).
template <bool DEFER, bool TRIM, uint32_t QCAP, bool SKIP, bool TRACK>
__device__ __forceinline__ uint32_t consume(const Pass1Args &a, const ColTables &t, const DictTable &d, const P1Lds &L, WaveState &ws, uint32_t cnt,
                                            uint32_t lane)
{
    // How this function is written.  What a lane IS (a line end, a field, a text field still looking for its slot ...) is kept as
    // a 64-bit LANE MASK in scalar registers, not as a bool per lane: masks come from ballots of single comparisons (a v_cmp that
    // writes the scalar pair directly), are combined on the scalar unit, and become the exec mask of a block with lane_in().
    // A bool that is changed inside a branch and looked at behind it costs a v_cndmask and a v_cmp per use instead (it lives in a
    // vector register as 0 / 1), and this kernel is bound by the vector ALU's issue rate (profiles/r03_valu_issue_rate.txt).
    constexpr uint32_t DB = desc_bytes(SKIP), QM = QCAP * DB - 1u;
    const uint64_t livem = mask_lt((int)cnt);
    const uint32_t di = ws.q_head + lane;
    // a descriptor: position of the boundary (24 bits, relative to the visit) | its ordinal among the visit's boundaries (8 bits).
    // Tabs that directly follow a boundary -- empty fields -- have no descriptor (the tokeniser drops them): the ordinals of two
    // neighbours tell how many were skipped between them.
    // (SKIP: the kernel for tables of mostly empty fields.  Without it a descriptor is the position alone, every boundary has one,
    // and everything about ordinals below folds away.)
    uint32_t dsc;
    if (SKIP) dsc = lds_u32(ws.desc | ((di << 2) & QM));
    else dsc = near_below(ws.pos_hi, lds_u16(ws.desc | ((di << 1) & QM)));      // (the position itself)
    const uint32_t e_pos = SKIP ? dsc & DESC_POS : dsc;
    // the boundary before mine: my left neighbour's, lane 0: the last one of the batch before (one DPP move, no second LDS read)
    uint32_t pdsc = (uint32_t)__builtin_amdgcn_update_dpp((int)(SKIP ? ws.last_e | (ws.last_ord << 24) : ws.last_e), (int)dsc, DPP_WAVE_SHR1, 0xf, 0xf, false);
    uint32_t pe_pos = SKIP ? pdsc & DESC_POS : pdsc;
    uint32_t n_prev = 0u;                                               // empty fields skipped between that boundary and my field
    if (SKIP) n_prev = lane_in(livem) ? ((dsc >> 24) - (pdsc >> 24) - 1u) & 0xffu : 0u;
    uint64_t farm = (SKIP && ws.far) ? 1ull : 0ull;                     // lanes whose field begins behind a boundary too far back for a descriptor
    const uint32_t d_last = lane_value(dsc, (int)cnt - 1);
    const uint32_t e_last = SKIP ? d_last & DESC_POS : d_last;
    // ---- what kind of boundary is mine?  The tokeniser only said "a tab or a newline": the byte itself and the one
    // ---- before it (a backslash?) come from the byte ring
    const uint32_t ea = ws.bytes + (e_pos & (BYTE_RING - 1u));
    const uint32_t be = zdw_lds[ea], bp = zdw_lds[ea - 1u];
    uint64_t sentm = ws.first_pending ? 1ull : 0ull;                    // the lane whose line end precedes the visit's first row (its byte may lie before anything the visit loaded)
    uint64_t nlm = (BAL(be == '\n') | sentm) & livem;                   // line ends that count
    uint64_t deadm = ~livem;             // takes no part: beyond cnt, escaped, foreign, past the end of the visit
    uint64_t vm = 0;                     // escaped separators of this batch
    // Anything out of the ordinary in this batch?  A backslash in front of a boundary, a boundary whose bytes the ring no
    // longer holds, a visit that has not found its first row yet, the end of the span: all rare, all in one branch.
    const uint64_t oddm = (BAL(bp == '\\') | BAL(e_pos <= ws.ring_lo)) & livem & ~sentm;
    if ((oddm != 0) | ws.ignoring | (e_last >= ws.rel_se1)) {
        if (oddm) {
            // exact: the byte and the parity of the backslash run before it, from the image (Q1, Q4: special iff an even
            // number of backslashes precedes)
            uint32_t esc = 0, nl = 0;
            if (lane_in(oddm)) {
                const uint64_t p = ws.vbase + e_pos;
                const uint8_t *img = COLD(in);
                nl = img[p] == '\n' ? 1u : 0u;
                esc = backslashes_before(img, COLD(lo), p) & 1u;
            }
            vm = BAL(esc != 0u);          // (the loads have landed before anything below looks at their values)
            nlm = (nlm & ~oddm) | (BAL(nl != 0u) & ~vm);
            deadm |= vm;
        }
        // a visit that began inside a row of the span before: that row's end is the line end before our first row
        if (ws.ignoring) {
            if (!nlm) { ws.last_e = e_last; ws.last_ord = d_last >> 24; ws.far = false; return 0; }                 // all of it foreign
            const int s = lowest_bit(nlm);
            sentm = 1ull << s;
            deadm |= mask_lt(s);
            ws.ignoring = false;
        }
        // the visit ends with the first line end at or past the last byte of its span
        const uint64_t endm = nlm & ~sentm & BAL(e_pos >= ws.rel_se1);
        if (endm) {
            const uint64_t after = ~mask_le(lowest_bit(endm));
            deadm |= after; nlm &= ~after;
            ws.done = true;
        }
    }
    uint32_t fs = pe_pos + 1u + n_prev;                                 // a field begins one byte after the boundary before it and the tabs dropped behind that
    uint32_t raw_len, len, row, col, rowd = 0;   // (rowd: rows that end in this batch before my field)
    const uint32_t row_first = ws.row_cur;
    uint64_t fieldm, inblkm;             // lanes that hold a field of the table / whose row belongs to the block
    if (nlm == 0 && vm == 0 && !ws.first_pending) {
        // The common batch: 64 fields of one row, nothing out of the ordinary -- no line end, so no row ends, no blank line, no
        // sentinel: the row is the wave's, the column counts on.  (Seven in ten batches of the benchmark's table, nineteen in
        // twenty of a table of mostly empty fields.)
        raw_len = len = e_pos - fs;
        row = ws.row_cur;
        if (SKIP) {
            const uint32_t incl = wave_incl_sum(lane_in(deadm) ? 0u : n_prev + 1u);      // columns: one per field, skipped ones included
            col = ws.col_cur + incl - 1u;
            ws.col_cur += lane_value(incl, 63);
        } else {
            col = ws.col_cur + lane;
            ws.col_cur += cnt;
        }
        const bool in_block = ws.row_cur < a.row_limit;                 // (one row: the same for every lane)
        inblkm = in_block ? ~0ull : 0ull;
        fieldm = ~deadm & BAL(col < a.ncols);
        if (!in_block) fieldm = ws.row_cur == a.row_limit ? fieldm & BAL(col < a.partial_fields) : 0ull;
        ws.prev_nl = false;
        ws.last_e = e_last; ws.last_ord = d_last >> 24;
        ws.far = false;
    } else {
    uint64_t pnlm = (nlm << 1) | (ws.prev_nl ? 1ull : 0ull);            // the boundary before mine is a line end
    uint64_t realm = 0;                                                 // (vm != 0 only) boundaries that count
    if (any_lane(vm)) {
        // rare: separators of this batch are escaped -- my field begins after the last boundary that counts
        realm = ~deadm;
        uint32_t pnl = lane_in(pnlm) ? 1u : 0u, farv = lane_in(farm) ? 1u : 0u;
        const uint64_t before = realm & mask_lt((int)lane);
        const bool dead = lane_in(deadm);
        if (!dead && before) {
            const int p = highest_bit(before);
            if ((uint32_t)p + 1u != lane) {
                if (SKIP) pdsc = lds_u32(ws.desc | (((ws.q_head + (uint32_t)p) << 2) & QM));
                else pdsc = near_below(ws.pos_hi, lds_u16(ws.desc | (((ws.q_head + (uint32_t)p) << 1) & QM)));
                pe_pos = SKIP ? pdsc & DESC_POS : pdsc;
                // (the descriptors between that one and mine are escaped separators: kept, no fields)
                if (SKIP) n_prev = ((dsc >> 24) - (pdsc >> 24) - 1u - (lane - (uint32_t)p - 1u)) & 0xffu;
                fs = pe_pos + 1u + n_prev;
                pnl = (uint32_t)(nlm >> p) & 1u;
                farv = 0u;
            }
        } else if (!dead && lane > 0) {                                 // none in this batch: the one before the batch
            pe_pos = ws.last_e;
            if (SKIP) n_prev = ((dsc >> 24) - ws.last_ord - 1u - lane) & 0xffu;
            fs = pe_pos + 1u + n_prev;
            pnl = ws.prev_nl ? 1u : 0u;
            farv = (SKIP && ws.far) ? 1u : 0u;
        }
        pnlm = BAL(pnl != 0u);
        if (SKIP) farm = BAL(farv != 0u);
    }
    if (SKIP) { if (lane_in(sentm)) n_prev = 0u; }
    raw_len = len = e_pos - fs;
    uint64_t blankm = nlm & pnlm & ~sentm & BAL(len == 0u);             // line end right after a line end: no row (getnextrow.cpp:38-43)
    if (SKIP) blankm &= BAL(n_prev == 0u);
    const uint64_t rm = nlm & ~blankm & ~sentm;                         // line ends that end a row
    // columns: every boundary that counts ends a field, and n_prev more (empty ones) lie in front of it; a line end starts over
    uint32_t incl = 0, incl_all = 0, incl_at_last_nl = 0;
    if (SKIP) {
        incl = wave_incl_sum(lane_in(deadm | sentm) ? 0u : n_prev + 1u);
        incl_all = lane_value(incl, 63);
    }
    row = ws.row_cur; col = SKIP ? ws.col_cur + incl - 1u : ws.col_cur + lane;
    if (any_lane(nlm)) {
        rowd = mbcnt64(rm);
        row += rowd;
        const uint64_t prev = nlm & mask_lt((int)lane);
        if (SKIP) {
            const uint32_t incl_prev_nl = (uint32_t)__shfl((int)incl, prev ? highest_bit(prev) : 0, 64);      // (every lane takes part in the exchange)
            if (prev) col = incl - incl_prev_nl - 1u;
            incl_at_last_nl = lane_value(incl, highest_bit(nlm));
        } else if (prev) {
            col = lane - 1u - (uint32_t)highest_bit(prev);
        }
        // rows of 16383 bytes or more can grow the reference's line buffer (m_LongestLine): the host replays exactly those.
        // Only a batch whose last line end lies that far past the start of the open row can hold one.
        const int h = highest_bit(nlm);
        const uint64_t abs_h = ws.vbase + lane_value(e_pos, h);
        if (abs_h + 1 - ws.row_start >= FIRST_ROW_CAP - 1) {
            const uint32_t e_prev = (uint32_t)__shfl((int)e_pos, prev ? highest_bit(prev) : 0, 64);      // (every lane takes part in the exchange)
            const uint64_t start = prev ? ws.vbase + e_prev + 1 : ws.row_start;
            const uint64_t rl = ws.vbase + e_pos + 1 - start;
            if (lane_in(nlm & ~sentm)) {
                if (rl >= FIRST_ROW_CAP - 1) {
                    const uint32_t li = atomicAdd(&COLD(st)->long_count, 1u);
                    if (li < COLD(long_cap)) { LongRow *lr = COLD(long_rows); lr[li].start = start; lr[li].len = rl; }
                }
            }
        }
        ws.row_start = abs_h + 1;
    }
    if (!SKIP && any_lane(vm)) {
        // columns count the boundaries that count
        const uint64_t before = realm & mask_lt((int)lane), prev = nlm & mask_lt((int)lane);
        col = prev ? popc64(before & ~mask_le(highest_bit(prev))) : ws.col_cur + popc64(before);
    }
    if (any_lane(rm)) {
        if (lane_in(rm)) {
            if (row <= a.row_limit) {
                if (col + 1 != a.ncols) atomicMin(&COLD(st)->bad_row, row);          // ConvertToZDW.cpp:336-337
                if (row + 1 == a.row_limit) COLD(st)->split_pos = ws.vbase + e_pos + 1;  // the next block starts after this row
            }
        }
    }
    inblkm = BAL(row < a.row_limit);
    fieldm = ~deadm & ~sentm & ~blankm & BAL(col < a.ncols) & (inblkm | (BAL(row == a.row_limit) & BAL(col < a.partial_fields)));
    // state for the next call
    ws.row_cur += popc64(rm);
    if (SKIP) ws.col_cur = nlm ? incl_all - incl_at_last_nl : ws.col_cur + incl_all;
    else if (any_lane(vm)) ws.col_cur = any_lane(nlm) ? popc64(realm & ~mask_le(highest_bit(nlm))) : ws.col_cur + popc64(realm);
    else ws.col_cur = any_lane(nlm) ? cnt - 1u - (uint32_t)highest_bit(nlm) : ws.col_cur + cnt;
    if (any_lane(vm)) {
        if (realm) {                     // the next batch's first field begins behind the last boundary that counted
            const int h = highest_bit(realm);
            ws.prev_nl = ((nlm >> h) & 1ull) != 0;
            ws.last_e = SKIP ? lane_value(dsc, h) & DESC_POS : lane_value(dsc, h);
            ws.far = false;
        }
        ws.last_ord = d_last >> 24;      // (escaped separators behind that boundary have descriptors and ordinals of their own: as if it were the last of them)
    } else {
        ws.prev_nl = ((nlm >> (cnt - 1u)) & 1ull) != 0;          // (lanes a foreign row or the end of the visit took out: nothing reads it then)
        ws.last_e = e_last; ws.last_ord = d_last >> 24;
        ws.far = false;
    }
    ws.first_pending = false;
    }
    if (P1_ABLATE(a, 1u)) fieldm = 0;

    // tables of mostly empty fields: the record matrix was zeroed beforehand (one streaming write instead of a scattered 4- or
    // 8-byte store per empty field), so an empty field -- no record but zero, no range, no string (ConvertToZDW.cpp:340-341) -- is done
    // (SKIP: the empty fields in front of mine have no descriptor of their own; the record matrix was zeroed before the launch)
    constexpr bool MAY_BE_ZEROED = SKIP || QCAP == QCAP_LARGE;      // (only the forms for tables of mostly empty fields run on a zeroed matrix: launch_pass1)
    if (MAY_BE_ZEROED && a.rec_zeroed) fieldm &= BAL(raw_len != 0u);
    // a field that begins further back than a descriptor can say (the visit moved its base past its start): the general form
    // (its absolute position is needed by new strings and general-form fields only: computed where they are handled)
    uint64_t far_start = 0;
    if (SKIP && farm) {
        if (lane_in(farm)) {
            far_start = ws.far_abs + 1u + n_prev;
            const uint64_t l64 = ws.vbase + e_pos - far_start;
            if (l64 >= (1ull << 30)) { COLD(st)->nul_seen = 2; }            // a single field of 1 GiB or more
            raw_len = len = (uint32_t)l64;
        }
    }
    auto abs_start_of = [&]() -> uint64_t { return (SKIP && lane_in(farm)) ? far_start : ws.vbase + fs; };
    uint32_t meta = 0;
    if (t.cols_in_lds) { if (lane_in(fieldm)) meta = lds_u32(t.meta + 4u * col); }
    const uint32_t kind = meta & 3u;
    // fields for the general form (a.trim == TRIM: launch_pass1 picks the kernel by it)
    uint64_t slowm = t.cols_in_lds ? fieldm & (BAL(fs < ws.ring_lo) | farm) : fieldm;
    const uint32_t ra = ws.bytes + (fs & (BYTE_RING - 1u));
    if (TRIM) {
        uint32_t general = 0;
        if (lane_in(fieldm & ~slowm & BAL(len != 0u))) {
            if (len > 32u) general = 1u;                  // beyond what the byte ring's tail copy lets this path read in one piece
            else {
                // blanks at the end of the field: its last 8 bytes (or what precedes it, never counted: k <= len)
                uint32_t tw[2];
                lds_bytes<2>(ra + len - 8u, tw);
                const uint64_t x = u64_of(tw[0], tw[1]) ^ 0x2020202020202020ull;
                const uint64_t f = (((x & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full) | x) & 0x8080808080808080ull;      // bit 7 of a byte: it is no blank
                uint32_t k = f ? (uint32_t)__builtin_clzll(f) >> 3 : 8u;
                if (k > len) k = len;
                if (k == 8u && len > 8u) general = 1u;      // there may be more of them: general form
                else len -= k;
            }
        }
        slowm |= BAL(general != 0u);
    }
    const uint64_t fastm = fieldm & ~slowm;
    // my field's record: the batch's first row (wave-uniform: scalar arithmetic) + a 32-bit offset, so the stores take the base from
    // scalar registers.  (The kernel that queues its general-form fields only runs with the column tables in LDS, which bounds the
    // stride far below 2^24; the other one serves tables of any width.)
    uint8_t *const rec_first = a.rec + (uint64_t)row_first * ws.rec_stride;
    const uint32_t rec_off = DEFER ? __umul24(rowd, ws.rec_stride & 0xffffffu) + (meta >> 16) : 0u;
    auto recp_of = [&]() -> uint8_t * { return DEFER ? rec_first + rec_off : a.rec + (uint64_t)row * ws.rec_stride + (meta >> 16); };
    uint64_t newm = 0;                   // lanes whose string was new to the dictionary

    // ---- text-like columns, <= 24 bytes: decided by one table slot.  The slot is REQUESTED here and looked at behind the integer fields
    // ---- of the batch (other lanes), whose records are stored behind that wait too: between the request and the wait the wave issues no
    // ---- other memory instruction, so the wait is for the probe alone (and for what was in flight long before it).
    const uint64_t textm = BAL(kind == K_TEXT);
    uint64_t ftm = fastm & textm;
    {
        const uint64_t longm = ftm & BAL(len > 24u);
        slowm |= longm; ftm &= ~longm;
    }
    uint32_t t_id = 0;
    uint64_t pendm = 0;                  // text fields that have not found their slot yet
    // (set and read by the lanes of wantm only: no zeroes moved into ten registers for the others)
    uint64_t c0, c1, c2, acc, slot_i;
    SlotLoad w0;
    // The column's flag in LDS: 1 = a string has been seen, 3 = ... and the workgroup's last probe for this column found an empty slot
    // (a column of mostly new strings: ids, time stamps).  Such a column's probes read at agent scope straight away
    // (request_slot_scoped): one access to the table less per new string.
    uint32_t col_flag = 0;
    uint64_t agentm = 0;
    if (ftm) {
        const uint64_t wantm = ftm & BAL(len != 0u);                              // empty fields are skipped (:340-341)
        if (lane_in(wantm)) {
            col_flag = zdw_lds[t.seen + col];
            if (!col_flag) zdw_lds[t.seen + col] = 1;
        }
#ifndef ZDW_P1_NO_SCOPE_HINT          // (A/B: tools/build_variant.sh)
        agentm = wantm & BAL((col_flag & 2u) != 0u);
#endif
        if (lane_in(wantm)) {
            const uint32_t lo = L.lut + 32u * len;
            uint32_t fw[6];
            lds_bytes<6>(ra, fw);
            c0 = u64_of(fw[0], fw[1]) & lds_u64(lo); c1 = u64_of(fw[2], fw[3]) & lds_u64(lo + 8); c2 = u64_of(fw[4], fw[5]) & lds_u64(lo + 16);
            acc = nh24(c0, c1, c2);
            slot_i = hash_place(acc) & a.slot_mask;
            if (!P1_ABLATE(a, 2u)) request_slot_scoped(a.slots + 4 * slot_i, w0, agentm);
        }
        if (!P1_ABLATE(a, 2u)) pendm = wantm;
    }

    // ---- integers of <= 16 plain digits, char(1) ----
    uint64_t fnm = fastm & ~textm;
    {
        const uint64_t bigm = fnm & BAL(kind == K_INT) & BAL(len > 16u);
        slowm |= bigm; fnm &= ~bigm;
    }
    uint64_t nstorem;                      // lanes with a numeric record to store (behind the text fields' wait)
    uint64_t n_rec = 0;                    // (an empty field's record)
    {
        const uint64_t emptym = fnm & BAL(len == 0u);     // an empty numeric field is a zero record and touches no range (sparse tables: most of them)
        nstorem = emptym & inblkm;
        fnm &= ~emptym;
    }
    if (fnm) {
        uint32_t junk = 0;
        if (lane_in(fnm)) {
            uint64_t v_rec = 0, v_range = 0;
            if (kind == K_INT) {
                // the last min(len, 8) characters and the (len - 8) before them, each right-aligned in 8 with '0' in front
                const uint32_t k = len > 8u ? len - 8u : 0u;
                const uint64_t m2 = lds_u64(L.lut + 32u * (8u + k - len)), m1 = lds_u64(L.lut + 32u * (8u - k));
                const uint64_t Z = 0x3030303030303030ull;
                // (with fewer than 8 characters the first word lies before the field and is masked whole)
                uint32_t nw[4];
                lds_bytes<4>(ra + len - 16u, nw);
                const uint64_t w2 = (u64_of(nw[2], nw[3]) & ~m2) | (Z & m2);
                const uint64_t w1 = (u64_of(nw[0], nw[1]) & ~m1) | (Z & m1);
                const uint32_t r0 = (uint32_t)w1, r1 = (uint32_t)(w1 >> 32), r2 = (uint32_t)w2, r3 = (uint32_t)(w2 >> 32);
                const uint32_t t0 = r0 - 0x30303030u, t1 = r1 - 0x30303030u, t2 = r2 - 0x30303030u, t3 = r3 - 0x30303030u;
                junk = nondigit_bits(r0, t0) | nondigit_bits(r1, t1) | nondigit_bits(r2, t2) | nondigit_bits(r3, t3);
                const uint32_t lo8 = __umul24(four_digits(t2), 10000u) + four_digits(t3);
                const uint32_t hi8 = __umul24(four_digits(t0), 10000u) + four_digits(t1);
                v_rec = v_range = (uint64_t)hi8 * 100000000ull + lo8;
            } else {   // char(1): pass 1 adds byte 1 only after a backslash (:358-361), pass 2 always (:543-547)
                uint32_t cw[1];
                lds_bytes<1>(ra, cw);
                const uint32_t w = cw[0];
                const uint32_t b0 = len ? (w & 0xffu) : 0u, b1 = len >= 2u ? ((w >> 8) & 0xffu) : 0u;
                const int64_t c0 = (int8_t)b0;
                const int64_t hi8 = (int64_t)((int)(int8_t)b1 * 256);
                v_range = (uint64_t)(b0 == '\\' ? c0 + hi8 : c0);
                v_rec = (uint64_t)(c0 + hi8);
            }
            if ((junk == 0u) & (v_range > 0)) range_update(a, t, col, (meta >> 2) & 0x3fffu, v_range);
            n_rec = v_rec;
        }
        const uint64_t junkm = BAL(junk != 0u);           // signs, blanks, letters: the general form
        slowm |= junkm;
        nstorem |= fnm & ~junkm & inblkm;
    }
    if (P1_ABLATE(a, 4u)) nstorem = 0;

    // ---- the text fields' slots have arrived (or will have, behind the wait) ----
    if (ftm) {
        // runs while any text lane is undecided; every lane of the wave passes through (ids are handed out at this level)
        for (uint32_t probe = 0; probe < 4; ++probe) {
            if (!pendm) break;
            uint64_t *slot = a.slots + 4 * slot_i;
            SlotWords w;
            if (probe == 0) {
                w = await_slot(w0);                                               // (lanes that requested nothing hold nothing: every mask below lies within pendm)
                // that was an ordinary load, possibly from this CU's L1: only a complete slot can be taken at its word (request_slot)
                // (the lanes of agentm have read at agent scope already)
                const uint64_t againm = pendm & ~agentm & (BAL(w.key == 0) | BAL(slot_incomplete(w)));
                if (againm) { if (lane_in(againm)) w = load_slot(slot); }
            } else if (lane_in(pendm)) w = load_slot(slot);
            const uint64_t samem = pendm & BAL(w.s0 == c0) & BAL(w.s1 == c1) & BAL(w.s2 == c2) & BAL((((uint32_t)(w.key >> KEY_LEN_SHIFT)) & 0xffu) == len);
            if (lane_in(samem)) {
                t_id = (uint32_t)w.key;
                if (TRACK) note_occurrence(a.uniq_pos, t_id - 1u, abs_start_of());
            }
            pendm &= ~samem;
            const uint64_t freem = pendm & BAL(w.key == 0);                       // an empty slot: the string is new (inserts are rare: one string in eighty of the benchmark's table)
            if (probe == 0) {                                                     // the column's flag follows what its probes find
                if (lane_in(samem & agentm)) zdw_lds[t.seen + col] = 1;
                if (lane_in(freem & ~agentm)) zdw_lds[t.seen + col] = 3;
            }
            if (freem) {
                const uint32_t nid = alloc_ids(ws.ids, freem, COLD(st), ID_CHUNK);
                uint32_t won = 0;                                                 // 1: the slot is ours, 2: another wave took it first
                if (lane_in(freem)) {
                    if (nid >= COLD(id_cap)) { COLD(st)->overflow = 1; }
                    else {
                        const uint64_t old = claim_slot<false>(slot, make_slot_key(hash_tag(acc), len, nid), nid, abs_start_of(), len, COLD(uniq_pos), COLD(uniq_len), c0, c1, c2, TRACK, COLD(uniq_key));
                        if (old == 0) { won = 1u; t_id = nid + 1; } else won = 2u;
                    }
                }
                newm |= BAL(won == 1u);
                slowm |= BAL(won == 2u);                                          // lost the race: general form
                pendm &= ~freem;
            }
            const uint64_t halfm = pendm & BAL(slot_incomplete(w));               // claimed, inline copy not complete yet: general form
            slowm |= halfm; pendm &= ~halfm;
            if (lane_in(pendm)) slot_i = (slot_i + 1) & a.slot_mask;              // somebody else's: the next slot
        }
        slowm |= pendm;
        uint64_t tstorem = ftm & ~slowm & inblkm;
        if (P1_ABLATE(a, 4u)) tstorem = 0;
        if (lane_in(tstorem)) {
#ifdef ZDW_P1_NT_REC
            __builtin_nontemporal_store(t_id, reinterpret_cast<uint32_t *>(recp_of()));
#else
            *reinterpret_cast<uint32_t *>(recp_of()) = t_id;
#endif
            if (SKIP && a.nz && t_id) atomicOr(a.nz + (uint64_t)row * a.nz_stride + (col >> 5), 1u << (col & 31u));
        }
    }
    // the integer fields' records
    if (nstorem) {
        if (lane_in(nstorem)) {
#ifdef ZDW_P1_NT_REC
            __builtin_nontemporal_store(n_rec, reinterpret_cast<uint64_t *>(recp_of()));
#else
            *reinterpret_cast<uint64_t *>(recp_of()) = n_rec;
#endif
            if (SKIP && a.nz && n_rec) atomicOr(a.nz + (uint64_t)row * a.nz_stride + (col >> 5), 1u << (col & 31u));
        }
    }

    // ---- everything else: the general form, from global memory ----
    const uint64_t sm = slowm;
    if (sm) {
        if (DEFER) {
            // Queue entries come from the wave's chunk (one returning atomic on the queue counter per call would bound a
            // block full of such fields: a single address takes about 30 M of them a second).  What is left of a chunk
            // that cannot take this call's fields is filled with blanks (length field all ones), which the second kernel skips.
            const uint32_t k = popc64(sm);
            const uint32_t qs = L.dqs + ws.desc / (QCAP * DB) * 8u;          // ws.desc = wave index x ring size
            uint32_t cur = lds_u32(qs), end = lds_u32(qs + 4u);
            if (end - cur < k) {
                const uint32_t left = end - cur;
                uint64_t *dq0 = COLD(dq);
                if (lane < left && cur + lane < COLD(dq_cap)) dq0[2ull * (cur + lane) + 1] = ~0ull;
                uint32_t fresh = 0;
                if (lane == 0) fresh = atomicAdd(&COLD(st)->deferred, DQ_CHUNK);
                cur = lane_value(fresh, 0);
                end = cur + DQ_CHUNK;
                if (lane == 0) *reinterpret_cast<uint32_t *>(zdw_lds + qs + 4u) = end;
            }
            const uint32_t qb = cur + mbcnt64(sm);
            if (lane == 0) *reinterpret_cast<uint32_t *>(zdw_lds + qs) = cur + k;
            {
                uint64_t *dq0 = COLD(dq);
                const uint32_t dqc = COLD(dq_cap);
                if (lane_in(sm)) {
                    if (qb < dqc) {
                        dq0[2ull * qb] = abs_start_of() | ((uint64_t)col << 40);
                        dq0[2ull * qb + 1] = (uint64_t)raw_len | ((uint64_t)row << 32);
                    } else {
                        COLD(st)->deferred_lost = 1;
                    }
                }
            }
        } else {
            bool is_new = false;
            if (lane_in(sm)) process_field_general(a, t, d, ws.ids, abs_start_of(), raw_len, col, row, lane_in(inblkm), is_new);
            adopt_ids(ws.ids, sm);
            newm |= __ballot(is_new);
        }
    }
    return popc64(newm);
}
// FOLD: no row scan has run -- the waves take the spans in order (a ticket each), count the rows that start in theirs and learn their
// first row's number from the spans before (Pass1Args.span_state).
// TRACK: uniq_pos[id] ends up as the string's FIRST occurrence (Pass1Args.track_first: every occurrence takes part in a minimum).
template <bool DEFER, bool TRIM, uint32_t QCAP, bool SKIP, bool FOLD, bool TRACK>
__global__ __launch_bounds__(64 * P1_WPB_MAX, DEFER ? P1_MIN_WAVES : 1) void k_pass1(const Pass1Args a)
{
    constexpr uint32_t DB = desc_bytes(SKIP), QM = QCAP * DB - 1u;
    // (the compiler cannot prove threadIdx.x >> 6 wave-uniform: without the readfirstlane everything derived from the wave's
    // number -- spans, positions, the whole visit bookkeeping -- lives in vector registers and every branch on it saves and
    // restores the exec mask)
    const uint32_t wv = uniform32(threadIdx.x >> 6);
    const uint32_t wpb = blockDim.x >> 6;                                        // waves of this workgroup (launch_pass1)
    const uint32_t wave_id = blockIdx.x * wpb + wv, wave_count = gridDim.x * wpb;
    const uint32_t lane = lane_id();
    // (the kernel that queues its general-form fields is only launched with the column tables in LDS: zdw_capi.hip)
    const bool cols_in_lds = DEFER ? true : a.use_lds_ranges != 0;
    const P1Lds L = p1_layout(a.ncols, a.nnum, cols_in_lds, QCAP * DB, wpb);
    ColTables t;
    t.meta = L.meta; t.mn = L.mn; t.mx = L.mx; t.seen = L.seen; t.cols_in_lds = cols_in_lds;

    // workgroup tables
    for (uint32_t i = threadIdx.x; i < LUT_ROWS * 8u; i += blockDim.x) {
        const uint32_t row = i >> 3, b0 = (i & 7u) * 4u;          // dword (i & 7) of row `row`: bytes b0 .. b0 + 3
        const uint32_t nb = row > b0 ? (row - b0 < 4u ? row - b0 : 4u) : 0u;
        *reinterpret_cast<uint32_t *>(zdw_lds + L.lut + 4u * i) = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
    }
    if (t.cols_in_lds) {
        for (uint32_t c = threadIdx.x; c < a.ncols; c += blockDim.x) {
            *reinterpret_cast<uint32_t *>(zdw_lds + t.meta + 4 * c) = pack_col_meta(COLD(col_kind)[c], (uint32_t)COLD(col_numidx)[c] & 0x3fffu, COLD(col_recoff)[c]);
            zdw_lds[t.seen + c] = 0;
        }
        for (uint32_t i = threadIdx.x; i < a.nnum; i += blockDim.x) {
            reinterpret_cast<uint64_t *>(zdw_lds + t.mn)[i] = ~0ull;
            reinterpret_cast<uint64_t *>(zdw_lds + t.mx)[i] = 0;
        }
    }
    __syncthreads();

    // Two arguments of the per-batch / per-round path that the compiler would otherwise read from the kernel-argument segment again
    // wherever it is short of scalar registers (a scalar load and an lgkmcnt(0) wait -- which also waits for the wave's LDS
    // operations -- in the middle of the loop): passed through an empty asm statement they are ordinary register values.
    uint32_t rebase_at = a.rebase_at, hot_stride = a.rec_stride;
    asm volatile("" : "+s"(rebase_at), "+s"(hot_stride));
    WaveState ws;
    ws.rec_stride = hot_stride;
    ws.desc = L.desc + wv * QCAP * DB;
    ws.bytes = L.bytes + wv * RING_STRIDE + RING_HEAD;
    ws.q_head = 0; ws.q_count = 0; ws.row_cur = 0; ws.col_cur = 0; ws.ring_lo = 0; ws.rel_se1 = 0; ws.first_pending = false; ws.vbase = 0;
    ws.ignoring = false; ws.prev_nl = false; ws.done = false; ws.row_start = 0; ws.last_e = 0;
    ws.last_ord = 255u; ws.tok_ord = 0; ws.tok_prevb = 0; ws.tok_bs = false; ws.far = false; ws.far_abs = 0; ws.pos_hi = 0;
    ws.ids.cur = ws.ids.end = ws.ids.taken = 0;
    if (lane < 2) *reinterpret_cast<uint32_t *>(zdw_lds + L.dqs + 8u * wv + 4u * lane) = 0;      // the wave's chunk of deferred-queue entries: [cur, end)
    DictTable d;
    d.slots = a.slots; d.slot_mask = a.slot_mask; d.uniq_pos = a.uniq_pos; d.uniq_len = a.uniq_len; d.id_cap = a.id_cap; d.uniq_key = a.uniq_key; d.st = a.st; d.track_first = TRACK ? 1u : 0u;
    uint32_t new_entries = 0;          // dictionary inserts of this wave not yet added to st->entries

    constexpr bool fold = FOLD;                     // no row scan has run: spans by ticket, their rows counted here
    for (uint32_t si = wave_id;; si += wave_count) {
        // what a span visit needs of the block's geometry: read per visit, not held across the kernel
        P1Status *const stp = COLD(st);
        const uint8_t *const v_in = COLD(in);
        const uint64_t v_lo = COLD(lo), v_hi = COLD(hi), v_span0 = COLD(span0);
        const uint32_t v_nspans = COLD(nspans), v_shift = COLD(span_shift);
        if (uniform32(*reinterpret_cast<volatile const uint32_t *>(&stp->overflow))) break;      // an ordinary (cached) read: a stale 0 only delays the exit
        if (new_entries >= 128) {        // publish this wave's inserts in bulk; stop everybody once the table is half full
            uint32_t seen_total = 0;
            if (lane == 0) seen_total = atomicAdd(&stp->entries, new_entries) + new_entries;
            seen_total = lane_value(seen_total, 0);          // (a readlane: wave-uniform for the compiler too, unlike a shuffle)
            new_entries = 0;
            if (seen_total >= COLD(max_entries)) { stp->overflow = 1; break; }          // (every lane stores: a branch on the lane number next to a break makes the loop's exit divergent for the compiler)
        }
        uint32_t rowbase, rows_here;
        if (fold) {
            // (nothing between taking a span and publishing its count may leave the loop: the spans behind it wait for that count)
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&stp->next_span, 1u);
            si = lane_value(t, 0);
            if (si >= v_nspans) break;
            uint64_t *const span_state = COLD(span_state);
            const uint64_t f0 = (v_span0 + si) << v_shift;
            const uint64_t fs0 = f0 > v_lo ? f0 : v_lo, fe0 = f0 + (1ull << v_shift) < v_hi ? f0 + (1ull << v_shift) : v_hi;
            uint32_t nul = 0;
            const uint32_t mine = count_row_starts(v_in, COLD(n), v_lo, fs0, fe0, lane, nul);
            (void)wave_excl_sum(mine, lane, rows_here);
            rows_here = uniform32(rows_here);
            if (__ballot(nul != 0)) stp->nul_seen = 1;
            uint64_t before = 0;                             // rows that start in the spans before this one
            if (si) {
                __hip_atomic_store(span_state + si, (1ull << 62) | rows_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // look back: 64 spans at a time, the nearest in lane 0; up to the nearest one that knows its total
                int64_t j = (int64_t)si - 1;
                uint32_t spins = 0;
                for (;;) {
                    const int64_t idx = j - (int64_t)lane;
                    uint64_t w = 2ull << 62;                  // (before span 0: a total of 0)
                    if (idx >= 0) w = __hip_atomic_load(span_state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t flag = (uint32_t)(w >> 62);
                    const uint64_t known = __ballot(flag == 2u), pending = __ballot(flag == 0u);
                    const int first = known ? lowest_bit(known) : 64;
                    if (pending & mask_lt(first)) {           // a nearer span has not counted yet
                        if (++spins > (1u << 22)) { stp->fold_abort = 1; stp->overflow = 1; break; }
                        __builtin_amdgcn_s_sleep(2);
                        continue;
                    }
                    uint32_t lo32 = (lane <= (uint32_t)first) ? (uint32_t)w : 0u, hi32 = (lane <= (uint32_t)first) ? (uint32_t)(w >> 32) & 0x3fffffffu : 0u;
                    uint32_t tl, th;
                    (void)wave_excl_sum(lo32 & 0xffffffu, lane, tl);                 // (sums of 64 values below 2^24 each stay below 2^32 ... piecewise)
                    uint32_t tm;
                    (void)wave_excl_sum(lo32 >> 24, lane, tm);
                    (void)wave_excl_sum(hi32, lane, th);
                    before += (uint64_t)uniform32(tl) + ((uint64_t)uniform32(tm) << 24) + ((uint64_t)uniform32(th) << 32);
                    if (first < 64) break;
                    j -= 64;
                }
            }
            __hip_atomic_store(span_state + si, (2ull << 62) | (before + rows_here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (si + 1 == v_nspans) stp->rows_total = before + rows_here + 1;
            if (before >= 0xFFFFFFFFull) break;              // (reported by the host: more rows than a block may hold)
            rowbase = (uint32_t)before;
        } else {
            if (si >= v_nspans) break;
            // (values loaded from memory count as divergent for the compiler even when the address is wave-uniform: a branch on
            // one would put the rest of the loop body under an exec mask and its bookkeeping into vector registers)
            const uint32_t *const span_rowbase = COLD(span_rowbase);
            rowbase = uniform32(span_rowbase[si]);
            rows_here = uniform32(span_rowbase[si + 1]) - rowbase;
        }
        if (rows_here == 0) continue;                             // no row starts in this span
        if (rowbase > a.row_limit) break;                         // later spans hold rows of later blocks only
        const uint64_t s0 = (v_span0 + si) << v_shift, SPAN_BYTES = 1ull << v_shift;
        const uint64_t ss = s0 > v_lo ? s0 : v_lo;
        const uint64_t se = s0 + SPAN_BYTES < v_hi ? s0 + SPAN_BYTES : v_hi;
        const uint64_t pos0 = ss & ~(uint64_t)(ROUND - 1);

        // the visit: all descriptors of the previous one were handled or dropped, the ring is empty
        const bool at_row_start = (ss == v_lo) || uniform32(is_row_terminator(v_in, v_lo, ss - 1) ? 1u : 0u) != 0;
        ws.vbase = (pos0 - ROUND) & ~(uint64_t)(BYTE_RING - 1);            // wraps below zero for the first span: only differences matter
        ws.q_head = 0; ws.q_count = 0;
        ws.row_cur = rowbase; ws.col_cur = 0;
        ws.first_pending = at_row_start;
        ws.ignoring = !at_row_start;         // the row open at ss began in an earlier span and belongs to it; its line end is our sentinel
        ws.prev_nl = false; ws.done = false;
        ws.row_start = ss;
        ws.last_e = (uint32_t)(ss - 1 - ws.vbase);
        ws.last_ord = 255u;                  // (the first boundary of the visit has ordinal 0)
        ws.tok_ord = 0; ws.tok_bs = false; ws.far = false; ws.far_abs = 0;
        ws.tok_prevb = at_row_start ? 1u : 0u;      // the byte before the visit's first row is a line end: tabs right at its start are empty fields
        if (at_row_start) {
            // (every lane stores the same word: a branch on the lane number here would make q_count divergent for the compiler)
            if (SKIP) *reinterpret_cast<uint32_t *>(zdw_lds + ws.desc) = (uint32_t)(ss - 1 - ws.vbase) | (255u << 24);
            else *reinterpret_cast<uint16_t *>(zdw_lds + ws.desc) = (uint16_t)(ss - 1 - ws.vbase);
            ws.q_count = 1;
        }

        // Per-round bookkeeping is 32-bit: positions relative to the visit base, clamped to [0, 2^31) (a visit
        // moves its base long before it gets there); 64-bit positions are rebuilt only on the rare paths.
        uint32_t rel_ss, rel_hi;
        auto rel31 = [&](uint64_t x) {
            const int64_t dlt = (int64_t)(x - ws.vbase);
            return dlt < 0 ? 0u : (dlt > 0x7fffffffll ? 0x7fffffffu : (uint32_t)dlt);
        };
        auto set_rels = [&]() { rel_ss = rel31(ss); ws.rel_se1 = rel31(se - 1); rel_hi = rel31(COLD(hi)); };
        set_rels();

        // this lane's kept boundaries `bits` (bit i = byte pos + i) from ring slot first_slot on; `all`: every boundary of the lane's
        // 16 bytes, ord0: the ordinal of the first of them
        auto emit = [&](uint32_t bits, uint32_t all, uint32_t ord0, uint32_t pos, uint32_t first_slot) {
            uint32_t qoff = first_slot * DB;
            while (bits) {
                const uint32_t i = (uint32_t)__builtin_ctz(bits);
                bits &= bits - 1;
                uint32_t dv = pos + i;
                if (SKIP) {
                    dv |= (ord0 + (uint32_t)__builtin_popcount(all & ((1u << i) - 1u))) << 24;
                    *reinterpret_cast<uint32_t *>(zdw_lds + (ws.desc | (qoff & QM))) = dv;
                } else {
                    *reinterpret_cast<uint16_t *>(zdw_lds + (ws.desc | (qoff & QM))) = (uint16_t)dv;
                }
                qoff += DB;
            }
        };

        // lanes whose chunk would start at or past the end of the image re-read the last 16-byte chunk that starts
        // inside it (an aligned chunk whose first byte is inside an allocation never leaves its page; their bytes lie
        // past `hi` and are masked)
        const uint8_t *last_chunk = v_in + ((COLD(n) - 1) & ~15ull);
        auto load_chunk = [&](const uint8_t *q) {
            q = q < last_chunk ? q : last_chunk;
#ifdef ZDW_P1_PLAIN_STREAM
            return *reinterpret_cast<const uint4 *>(__builtin_assume_aligned(q, 16));
#else
            // (read once: a non-temporal load, so that the image streaming through does not push the table's hot slots out of the CU's L1)
            typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(__builtin_assume_aligned(q, 16)));
            return make_uint4(x.x, x.y, x.z, x.w);
#endif
        };
        const uint8_t *lane_ptr = v_in + pos0 + 16u * lane;               // this lane's 16 bytes of chunk A of the current round
        uint4 P = load_chunk(lane_ptr);                                   // the one chunk in flight
        uint32_t relb = (uint32_t)(pos0 - ws.vbase);                      // relative position of the current round
        ws.pos_hi = relb + ROUND;
        uint32_t first_rel = relb;                                        // nothing below it was loaded by this visit
        ws.ring_lo = first_rel;
        // The visit as a small state machine around ONE place that handles descriptors (the field work is most of
        // this kernel's code; it is inlined once):
        //   step A      chunk A into the ring, chunk B requested; then the pending descriptors of the round before (all
        //               whole batches) while B travels
        //   step B      chunk B into the ring, chunk A of the next round requested; one prefix sum, the round's
        //               descriptors; then all but the last one or two batches (those wait for step A of the next round)
        //   step PIECE  a round denser than the ring: eight 16-lane pieces, the ring drained before each; a visit that
        //               has outgrown its relative positions moves its base here too
        //   step END    the image ended: whatever is pending
        enum : uint32_t { STEP_A = 0, STEP_B = 1, STEP_PIECE = 2, STEP_END = 3 };
        uint32_t step = relb < rel_hi ? STEP_A : STEP_END, piece = 0, piece_stop = 8;
        uint32_t mA = 0, mB = 0, ex2 = 0, tot2 = 0;          // kept boundaries of the round's chunks; exclusive prefix / total of their counts (A | B << 16)
        uint32_t bA = 0, bB = 0, exb2 = 0, totb2 = 0;        // all boundaries: the same
        uint32_t ord_round = 0;                              // ordinal of the round's first boundary
        bool rebase = false, near_end = false;
        for (;;) {
            uint32_t keep = 0;                    // descriptors that may stay pending after this step
            if (step <= STEP_B) {
                // a chunk: into the ring, the next one requested, its boundaries found.  (ONE place issues the load and ONE takes
                // its result: two would meet in a phi, and hipcc resolves that with register copies right behind the load --
                // a vmcnt(0) that waits for the chunk the moment it is requested)
                const bool second = step == STEP_B;
                const uint32_t at = (relb & (BYTE_RING - 1u)) + (second ? WROUND : 0u);     // 0, 1024, 2048 or 3072 (the base is a multiple of the ring)
                uint32_t mN, bN;
                {
                    const uint4 v = P;
                    *reinterpret_cast<uint4 *>(zdw_lds + ws.bytes + at + 16u * lane) = v;
                    if (at == 0 && lane < RING_TAIL / 16u) *reinterpret_cast<uint4 *>(zdw_lds + ws.bytes + BYTE_RING + 16u * lane) = v;
                    if (at == BYTE_RING - WROUND && lane == 63u) *reinterpret_cast<uint4 *>(zdw_lds + ws.bytes - RING_HEAD) = v;      // the ring's last 16 bytes, in front of it
                    // Every tab or newline is a boundary.  A tab that directly follows a boundary ends an EMPTY field: it gets no
                    // descriptor (the fields that do carry the ordinal of their boundary, so the lane that handles one sees how many
                    // were skipped in front of it) -- unless a backslash is near (this chunk or the one before): whether a boundary
                    // is escaped is only decided by the lane that handles it, so there every boundary keeps its descriptor.  Every
                    // 128th byte keeps its descriptor too: the ordinals of two neighbours then differ by less than 256.
                    const uint32_t m0 = tab_or_nl7(v.x), m1 = tab_or_nl7(v.y), m2 = tab_or_nl7(v.z), m3 = tab_or_nl7(v.w);
                    bN = pack16(m0, m1, m2, m3);
                    mN = bN;
                    if (SKIP) {      // (tables of mostly empty fields; for others the bookkeeping costs more than the lanes it saves)
                        const uint32_t n0 = m0 & (v.x << 6), n1 = m1 & (v.y << 6), n2 = m2 & (v.z << 6), n3 = m3 & (v.w << 6);      // 0x0a, not 0x09: bit 1
                        uint32_t nl16 = 0;
                        if (n0 | n1 | n2 | n3) nl16 = pack16(n0, n1, n2, n3);
                        const uint32_t x0 = v.x ^ 0x5c5c5c5cu, x1 = v.y ^ 0x5c5c5c5cu, x2 = v.z ^ 0x5c5c5c5cu, x3 = v.w ^ 0x5c5c5c5cu;
                        const uint32_t hz = (((x0 - 0x01010101u) & ~x0) | ((x1 - 0x01010101u) & ~x1) | ((x2 - 0x01010101u) & ~x2) | ((x3 - 0x01010101u) & ~x3)) & 0x80808080u;
                        const bool bs_here = __ballot(hz != 0u) != 0ull;
                        const uint32_t prevbit = (uint32_t)__builtin_amdgcn_update_dpp((int)ws.tok_prevb, (int)(bN >> 15), DPP_WAVE_SHR1, 0xf, 0xf, false);
                        uint32_t drop = bN & ~nl16 & ((bN << 1) | prevbit) & 0xffffu;
                        if ((lane & 7u) == 0u) drop &= ~1u;
                        if (bs_here | ws.tok_bs) drop = 0u;
                        mN = bN & ~drop;
                        ws.tok_bs = bs_here;
                        ws.tok_prevb = lane_value(bN >> 15, 63);
                    }
                }
                // (the chunk's registers are dead here: the next one is loaded into the same ones.  Requested while its
                // predecessor is still looked at it would land in other registers and be copied over at the loop's end,
                // behind a wait for it)
                uint32_t ahead = second ? ROUND : WROUND;
                asm volatile("" : "+v"(mN), "+v"(bN), "+v"(ahead));      // (an empty statement both values pass through: the boundaries are found before the address of the load exists)
                P = load_chunk(lane_ptr + ahead);
                constexpr bool DENSE = QCAP == QCAP_LARGE;      // boundaries of a chunk are emitted right away (two prefix sums per round)
                if (!second) ws.ring_lo = relb + WROUND > BYTE_RING + first_rel ? relb + WROUND - BYTE_RING : first_rel;      // the ring holds [relb - 3072, relb + 1024)
                else ws.ring_lo = relb > ROUND + first_rel ? relb - ROUND : first_rel;                                         // ... now [relb - 2048, relb + 2048)
                if (!second && !DENSE) {
                    mA = mN; bA = bN;
                    keep = P1_KEEP_AFTER_A;
                    step = STEP_B;
                } else {
                if (second) { mB = mN; bB = bN; if (DENSE) { mA = 0; bA = 0; } } else { mA = mN; bA = bN; mB = 0; bB = 0; }
                const uint32_t lrel = relb + 16u * lane;                                      // relative position of my first byte of chunk A
                // positions that exist for this visit: [ss, hi)
                if (relb < rel_ss || relb + ROUND > rel_hi) {
                    const uint32_t va = valid16r(lrel, rel_ss, rel_hi), vb = valid16r(lrel + WROUND, rel_ss, rel_hi);
                    mA &= va; bA &= va;
                    mB &= vb; bB &= vb;
                }
                // one prefix sum for both chunks; descriptors of A, then of B.  The ring always keeps the head's
                // predecessor, so it can take QCAP - 1 - pending new descriptors.
                static_assert(QCAP >= 256 + 2, "a drained ring must take one 16-lane piece");
                const uint32_t cnts = (uint32_t)__builtin_popcount(mA) | ((uint32_t)__builtin_popcount(mB) << 16);
                ex2 = wave_excl_sum(cnts, lane, tot2);
                if (SKIP) {
                    const uint32_t cntb = (uint32_t)__builtin_popcount(bA) | ((uint32_t)__builtin_popcount(bB) << 16);
                    exb2 = wave_excl_sum(cntb, lane, totb2);
                } else {
                    exb2 = ex2; totb2 = tot2;          // every boundary has its descriptor
                }
                const uint32_t totA = tot2 & 0xffffu, nb_total = totA + (tot2 >> 16);
                ord_round = ws.tok_ord;
                const uint32_t ordA = ord_round + (exb2 & 0xffffu), ordB = ord_round + (totb2 & 0xffffu) + (exb2 >> 16);      // ordinals of this lane's first boundaries
                ws.tok_ord += (totb2 & 0xffffu) + (totb2 >> 16);
                near_end = relb + ROUND > ws.rel_se1 || relb + ROUND >= rel_hi;          // the span ends in this round or has ended: find the visit's end promptly
                // (16-bit descriptors: what is still pending must lie within 64 KiB of this round's end -- it lies behind ws.last_e)
                const bool in_reach = SKIP || ws.q_count == 0 || relb + ROUND - uniform32(ws.last_e) <= 65000u;      // (last_e lives in a vector register: it feeds a DPP move)
                if (relb < rebase_at && ws.q_count + nb_total <= QCAP - 1 && in_reach) {
                    // the common round: everything fits
                    ws.pos_hi = relb + ROUND;
                    const uint32_t tail = ws.q_head + ws.q_count;
                    emit(mA, bA, ordA, lrel, tail + (ex2 & 0xffffu));
                    emit(mB, bB, ordB, lrel + WROUND, tail + totA + (ex2 >> 16));
                    ws.q_count += nb_total;
                    keep = near_end ? 0u : P1_KEEP_AFTER_B;
                    if (second) {
                        relb += ROUND; lane_ptr += ROUND;
                        step = relb < rel_hi ? STEP_A : STEP_END;
                    } else {
                        step = STEP_B;
                    }
                } else {
                    rebase = relb >= rebase_at;                                               // relative positions must stay below 2^31
                    piece = second ? (DENSE ? 4u : 0u) : 0u;                                  // (dense: the other chunk's mask is empty)
                    piece_stop = second ? 8u : 4u;
                    step = STEP_PIECE;                                                        // (first: everything that is pending)
                }
                }
            } else if (step == STEP_PIECE) {
                if (rebase) {
                    // nothing is pending: move the visit base up; the boundary before the head moves with it
                    const uint32_t delta = (relb - 2u * BYTE_RING) & ~(BYTE_RING - 1u);
                    if (!ws.first_pending) {
                        // (a field that began further back than the new base: its start is kept as an absolute position)
                        if (ws.last_e < delta) {
                            if (!SKIP) { if (lane == 0) COLD(st)->nul_seen = 2; }         // a single field of 1 GiB or more
                            else if (!ws.far) { ws.far = true; ws.far_abs = ws.vbase + ws.last_e; }
                            ws.last_e = delta;
                        }
                        ws.last_e -= delta;
                    }
                    ws.vbase += delta; ws.ring_lo -= delta; relb -= delta;
                    first_rel = 0;
                    set_rels();
                    rebase = false;
                }
                const bool inB = piece >= 4u;
                const uint32_t first_lane = 16u * (piece & 3u);
                const uint32_t ex = inB ? ex2 >> 16 : ex2 & 0xffffu, tot = inB ? tot2 >> 16 : tot2 & 0xffffu;
                const uint32_t piece_base = lane_value(ex, (int)first_lane);
                const uint32_t piece_end = first_lane == 48u ? tot : lane_value(ex, (int)first_lane + 16);
                const bool active = (lane >> 4) == (piece & 3u);
                const uint32_t lrel = relb + 16u * lane;
                ws.pos_hi = relb + ROUND;
                emit(active ? (inB ? mB : mA) : 0u, inB ? bB : bA, inB ? ord_round + (totb2 & 0xffffu) + (exb2 >> 16) : ord_round + (exb2 & 0xffffu),
                     inB ? lrel + WROUND : lrel, ws.q_head + ws.q_count + ex - piece_base);
                ws.q_count += piece_end - piece_base;
                if (++piece == piece_stop) {
                    keep = near_end ? 0u : P1_KEEP_AFTER_B;
                    if (piece_stop == 8u) {
                        relb += ROUND; lane_ptr += ROUND;
                        step = relb < rel_hi ? STEP_A : STEP_END;
                    } else {
                        step = STEP_B;                                                        // (dense: chunk A's pieces are out, on to chunk B)
                    }
                }
            }
            // ---- the one place descriptors are handled: 64 at a time while more than `keep` are pending ----
            const bool last = step == STEP_END;
            __builtin_amdgcn_wave_barrier();      // one wave per worker: LDS operations of a wave execute in order
            while (!ws.done && ws.q_count > keep) {
                const uint32_t cnt = ws.q_count < 64u ? ws.q_count : 64u;
                if (P1_ABLATE(a, 32u)) { if (near_end && ws.q_count <= 64u) ws.done = true; }      // profiling only: the tokeniser alone
                else new_entries += consume<DEFER, TRIM, QCAP, SKIP, TRACK>(a, t, d, L, ws, cnt, lane);
                ws.q_head += cnt;
                ws.q_count -= cnt;
            }
            __builtin_amdgcn_wave_barrier();
            if (ws.done || last) break;
        }
    }
    retire_ids(ws.ids, COLD(uniq_len), COLD(id_cap));
    if (DEFER) {
        const uint32_t qs = L.dqs + ws.desc / (QCAP * DB) * 8u;
        uint64_t *dq0 = COLD(dq);
        const uint32_t dqc = COLD(dq_cap);
        for (uint32_t q = lds_u32(qs) + lane; q < lds_u32(qs + 4u); q += 64)
            if (q < dqc) dq0[2ull * q + 1] = ~0ull;
    }
    __syncthreads();
    if (new_entries && lane == 0) {
        P1Status *stp = COLD(st);
        const uint32_t total = atomicAdd(&stp->entries, new_entries) + new_entries;
        if (total > COLD(max_entries)) stp->overflow = 1;
    }

    if (t.cols_in_lds) {   // one flush per workgroup
        const uint8_t *ck = COLD(col_kind);
        const int32_t *cn = COLD(col_numidx);
        uint8_t *cs = COLD(col_seen);
        uint64_t *cmn = COLD(col_min), *cmx = COLD(col_max);
        for (uint32_t c = threadIdx.x; c < a.ncols; c += blockDim.x) {
            if (ck[c] == K_TEXT) {
                if (zdw_lds[t.seen + c]) cs[c] = 1;
            } else {
                const uint32_t i = (uint32_t)cn[c];
                const uint64_t mn = reinterpret_cast<const uint64_t *>(zdw_lds + t.mn)[i], mx = reinterpret_cast<const uint64_t *>(zdw_lds + t.mx)[i];
                if (mx != 0) {
                    atomicMin(reinterpret_cast<unsigned long long *>(&cmn[c]), (unsigned long long)mn);
                    atomicMax(reinterpret_cast<unsigned long long *>(&cmx[c]), (unsigned long long)mx);
                }
            }
        }
    }
}

// The fields the fast kernel queued: one per lane through the general form, with the column tables and
// ranges in global memory (the fast kernel has flushed its own by then).
__global__ __launch_bounds__(256) void k_pass1_deferred(const Pass1Args a, uint32_t count)
{
    if (count == 0xFFFFFFFFu) {
        // launched right behind the main kernel: how many fields it queued is in the status block -- and nothing is to be done when
        // the queue or the table overflowed (the host runs the pass again)
        const P1Status *st = a.st;
        if (uniform32(st->deferred_lost | st->overflow)) return;
        count = uniform32(st->deferred);
        if (count > a.dq_cap) count = a.dq_cap;
    }
    const uint32_t lane = lane_id();
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = (gridDim.x * blockDim.x) >> 6;
    ColTables t;
    t.meta = t.mn = t.mx = t.seen = 0; t.cols_in_lds = false;
    DictTable d;
    d.slots = a.slots; d.slot_mask = a.slot_mask; d.uniq_pos = a.uniq_pos; d.uniq_len = a.uniq_len; d.id_cap = a.id_cap; d.uniq_key = a.uniq_key; d.st = a.st; d.track_first = a.track_first;
    IdAlloc al;
    al.cur = al.end = al.taken = 0;
    uint32_t created = 0;
    for (uint32_t g = wave * 64; g < count; g += nwaves * 64) {
        const uint32_t i = g + lane;
        bool is_new = false;
        bool mine = i < count;
        uint64_t w0 = 0, w1 = ~0ull;
        if (mine) { w0 = a.dq[2ull * i]; w1 = a.dq[2ull * i + 1]; }
        mine = mine && w1 != ~0ull;            // blanks: the unused tail of a wave's chunk
        const uint64_t mm = __ballot(mine);
        if (mine) {
            const uint32_t row = (uint32_t)(w1 >> 32);
            process_field_general<true>(a, t, d, al, w0 & POS_MASK, (uint32_t)w1, (uint32_t)(w0 >> 40), row, row < a.row_limit, is_new);
        }
        if (mm) adopt_ids(al, mm);
        created += popc64(__ballot(is_new));
    }
    retire_ids(al, a.uniq_len, a.id_cap);
    if (created && lane == 0) {
        const uint32_t total = atomicAdd(&a.st->entries, created) + created;
        if (total > a.max_entries) a.st->overflow = 1;
    }
}

// rec[row][column] := 0 wherever bit `column` of the map's row is clear
__global__ __launch_bounds__(256) void k_rec_zero_fill(uint8_t *__restrict__ rec, uint32_t rec_stride, uint32_t nrows, uint32_t ncols,
                                                       const uint8_t *__restrict__ col_kind, const uint32_t *__restrict__ col_recoff,
                                                       const uint32_t *__restrict__ nz, uint32_t nz_stride)
{
    for (uint32_t row = blockIdx.x; row < nrows; row += gridDim.x) {
        const uint32_t *m = nz + (uint64_t)row * nz_stride;
        uint8_t *r = rec + (uint64_t)row * rec_stride;
        for (uint32_t c = threadIdx.x; c < ncols; c += blockDim.x) {
            if ((m[c >> 5] >> (c & 31u)) & 1u) continue;
            if (col_kind[c] == K_TEXT) *reinterpret_cast<uint32_t *>(r + col_recoff[c]) = 0u;
            else *reinterpret_cast<uint64_t *>(r + col_recoff[c]) = 0ull;
        }
    }
}
void launch_rec_zero_fill(uint8_t *rec, uint32_t rec_stride, uint32_t nrows, uint32_t ncols, const uint8_t *col_kind, const uint32_t *col_recoff,
                          const uint32_t *nz, uint32_t nz_stride, hipStream_t s)
{
    if (!nrows || !ncols) return;
    hipLaunchKernelGGL(k_rec_zero_fill, dim3(nrows < 65536u ? nrows : 65536u), dim3(256), 0, s, rec, rec_stride, nrows, ncols, col_kind, col_recoff, nz, nz_stride);
}

void launch_pass1_deferred(const Pass1Args &a, uint32_t count, hipStream_t s)
{
    if (!count) return;
    const uint32_t want = count == 0xFFFFFFFFu ? 2048u : (count + 255) / 256;          // (all ones: the kernel reads the count on the device)
    hipLaunchKernelGGL(k_pass1_deferred, dim3(want < 2048u ? want : 2048u), dim3(256), 0, s, a, count);
}
void launch_pass1(const Pass1Args &a, hipStream_t s)
{
    if (!a.nspans) return;
    const bool defer = a.dq != nullptr;
    const bool trim = a.trim != 0;
    // three forms of the kernel: the ordinary one; a descriptor ring of 1024 for inputs with many boundaries per round; and, for
    // tables of mostly empty fields, the one whose tokeniser gives the tabs of empty fields no descriptors (few are left: the small ring)
    const uint32_t mode = a.skip_empty ? 2u : (a.dense_boundaries ? 1u : 0u);
    const uint32_t qcap = (mode == 1u ? QCAP_LARGE : QCAP_SMALL) * desc_bytes(mode == 2u);      // bytes of a wave's descriptor ring
    // Waves walk the spans with a grid stride, so every wave carries the same share of the block: the grid
    // must not exceed what is resident at once, or the surplus waves run alone after the others finish.
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    typedef void (*P1Fn)(const Pass1Args);
#define P1_MODES(D, T, F, K) {k_pass1<D, T, QCAP_SMALL, false, F, K>, k_pass1<D, T, QCAP_LARGE, false, F, K>, k_pass1<D, T, QCAP_SMALL, true, F, K>}
#define P1_FORMS(F, K) {{P1_MODES(false, false, F, K), P1_MODES(false, true, F, K)}, {P1_MODES(true, false, F, K), P1_MODES(true, true, F, K)}}
    static const P1Fn fns[2][2][2][2][3] = {{P1_FORMS(false, false), P1_FORMS(true, false)}, {P1_FORMS(false, true), P1_FORMS(true, true)}};
#undef P1_FORMS
#undef P1_MODES
    const P1Fn kfn = fns[a.track_first != 0][a.span_state != nullptr][defer][trim][mode];
    const void *fn = (const void *)kfn;
    // How many waves share a workgroup's column tables: the rings take ~6 KiB of LDS per wave and the tables their size once per
    // workgroup, so the number of waves a CU holds depends on how they are grouped -- 8 per workgroup for the benchmark's 200 columns
    // (3 workgroups: 24 waves, which is also what the registers allow), 10 for a schema of 1000 columns (2 x 10 instead of 2 x 8).
    // The choice is remembered per kernel and LDS layout.
    struct Choice { const void *fn; uint32_t ncols, nnum, lds_tables; uint32_t wpb, per_cu; };
    static thread_local Choice cache[8];
    static thread_local uint32_t cache_n = 0;
    uint32_t wpb = 0, per_cu = 0;
    const char *forced = test_hook("ZDW_P1_WPB");          // tests: every grouping (not remembered)
    for (uint32_t i = 0; !forced && i < cache_n && i < 8; ++i)
        if (cache[i].fn == fn && cache[i].ncols == a.ncols && cache[i].nnum == a.nnum && cache[i].lds_tables == a.use_lds_ranges) { wpb = cache[i].wpb; per_cu = cache[i].per_cu; }
    if (!wpb) {
        uint32_t best_waves = 0;
        for (uint32_t w = 4; w <= P1_WPB_MAX; w += 4) {          // (whole multiples of the CU's four SIMDs: 7 waves per workgroup, which would fit a 28th wave, ran 8 % slower than 8)
            const size_t l = p1_layout(a.ncols, a.nnum, a.use_lds_ranges != 0, qcap, w).total;
            if (l > 160u * 1024u) break;
            int pc = 0;
            // (more than 64 KiB of dynamic LDS per workgroup has to be asked for)
            if (l > 64u * 1024u && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l) != hipSuccess) { (void)hipGetLastError(); break; }
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&pc, fn, 64 * (int)w, l) != hipSuccess || pc < 1) { (void)hipGetLastError(); continue; }
            const size_t by_lds = (160u * 1024u) / (((l + 1279) / 1280) * 1280);     // LDS is granted in 1280-byte granules
            if ((size_t)pc > by_lds) pc = (int)by_lds;
            if (pc > 16) pc = 16;                                                     // workgroup slots per CU
            if (pc < 1) continue;
            const uint32_t waves = (uint32_t)pc * w;
            // (more waves; at equal numbers the default grouping, then the smaller workgroup)
            if (waves > best_waves || (waves == best_waves && w == P1_WPB)) { best_waves = waves; wpb = w; per_cu = (uint32_t)pc; }
        }
        if (!wpb) { wpb = P1_WPB; per_cu = 1; }
        if (forced) { wpb = (uint32_t)atoi(forced); if (wpb < 1 || wpb > P1_WPB_MAX) wpb = P1_WPB; per_cu = 1; }
        else cache[cache_n++ & 7u] = Choice{fn, a.ncols, a.nnum, a.use_lds_ranges, wpb, per_cu};
    }
    size_t lds = p1_layout(a.ncols, a.nnum, a.use_lds_ranges != 0, qcap, wpb).total;
    while (lds > 160u * 1024u && wpb > 1) lds = p1_layout(a.ncols, a.nnum, a.use_lds_ranges != 0, qcap, --wpb).total;      // (a forced grouping the schema's tables leave no room for)
    uint32_t max_blocks = (uint32_t)cus * per_cu;
    if (const char *e = experiment("ZDW_P1_WAVES_PER_CU")) max_blocks = (uint32_t)cus * (uint32_t)atoi(e) / wpb;   // tuning experiments
    const uint32_t want_blocks = (a.nspans + wpb - 1) / wpb;
    const uint32_t grid = want_blocks < max_blocks ? want_blocks : (max_blocks ? max_blocks : 1u);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(64 * wpb), lds, s, a);
}

}  // namespace zdw
