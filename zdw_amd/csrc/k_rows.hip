// k_rows.hip — pass 1 of the ZDW block encode on gfx950 in its "a lane walks a row" form: a streaming row index, then the
// tokenise / dedupe / parse kernel in which every lane of a wave handles its OWN rows, field by field, all lanes at the same column.
//
// Replaces the same reference code as k_pass1.hip: the per-row loop of ConvertToZDW::parseInput (cplusplus/ConvertToZDW.cpp:329-414),
// GetNextRow (getnextrow.cpp:26-84), GetDataRow / get_next_column (ConvertToZDW.cpp:265-323,1048-1067), Dictionary::insert
// (dictionary.cpp:31-51).  The two forms write the same records, ranges, dictionary table and status; zdw_capi.hip picks one per block.
//
// Why a second form.  k_pass1.hip hands 64 CONSECUTIVE FIELDS to the 64 lanes of a wave: loads coalesce perfectly, but finding the
// fields and handing them to lanes (a tokeniser that lists every boundary, descriptor rings, ballots for rows and columns) is 60 % of
// its vector instructions and 70 % of its scalar ones, and a batch mixes text and integer columns, so both code paths run for every
// batch at half occupancy.  That kernel is bound by instruction issue.  Here the work is transposed: a lane owns a
// row, the wave steps through the columns, so
//   * the column, its kind, its record offset and every branch on them are wave-uniform (scalar registers, scalar branches);
//   * a step is ONE kind of field for all 64 lanes (text: mask, hash, probe; integer: digits, range), nothing is executed for the
//     other kind;
//   * there is no list of boundaries: a lane finds the end of ITS field in a 28-byte window at its own cursor (one SWAR test per
//     dword, the flags packed by v_dot4, one find-first-bit), and the next field begins behind it;
//   * escapes (Q1, Q4) cannot straddle lanes: rows are found exactly by the row index kernel, which also says which 16 KiB spans hold
//     a backslash at all; a row that touches such a span is walked byte by byte by its lane and its fields go to the general form.
// What it costs: rows lie ~2 KB apart, so a wave's 64 cursors touch 64 different cache lines per step.  Each lane therefore keeps a
// ring of its row's bytes in LDS (dword-transposed: dword j of lane l at (j * 64 + l) * 4, conflict-free for any mix of j), refilled
// 64 bytes at a time by its own 16-byte loads well ahead of need; the bytes are classified once, when they arrive (a boundary bitmap
// beside the ring); records are staged in LDS per row and written out as whole 16-byte pieces by neighbouring lanes (a 4- or 8-byte
// store per lane and step to 64 different lines would be bound by the L2's request rate: 250 G/s, tools/probe_rate.hip).
//
// STATUS (round 4): exact -- every reference vector, the fuzz campaign -- and SLOWER than k_pass1.hip (10 GB: 16.0 ms +
// 4.0 ms for the row index against 10.8 ms), so zdw_capi.hip runs it only when asked to (ZDW_P1_ROWS).  The LDS it needs (26 KiB per
// wave) leaves 6 waves per CU; at 1.5 waves per SIMD the ~300 instructions of a step issue one every four to five cycles and every
// dependent LDS round trip and memory wait shows, and half of all text steps hold a lane whose slot needs a second look or an insert,
// which is handled synchronously.  What it would take is in DESIGN.md; what it proved useful for is there too.
#include "zdw_dev.h"
#include "zdw_knobs.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"
#include "zdw_field_dev.h"

#include <cstddef>
#include <cstdlib>
#include <type_traits>

namespace zdw {

// ------------------------------------------------------------------------------------------
// shared byte classes
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t rx_nz7b(uint32_t x) { return ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x; }      // bit 7 of a byte: it is non-zero (exact)
// bit 7 of each byte of w: the byte is a tab (0x09) or a newline (0x0a): y = byte ^ 8 maps them to 1 and 2; of the low seven bits m,
// m + 0x7f carries into bit 7 iff m >= 1 and m + 0x7d iff m >= 3; bytes >= 0x80 are out
__device__ __forceinline__ uint32_t rx_tab_or_nl7(uint32_t w)
{
    const uint32_t y = w ^ 0x08080808u, m = y & 0x7f7f7f7fu;
    return (m + 0x7f7f7f7fu) & ~(m + 0x7d7d7d7du) & ~y & 0x80808080u;
}
// flags in bit 7 of the bytes of a, b (0x80 or 0) -> one byte, bit i = flag of byte i (a: bits 0-3, b: bits 4-7), times 128:
// two byte-wise dot products (v_dot4_u32_u8) instead of shifts, masks and a multiply per dword
__device__ __forceinline__ uint32_t rx_pack8x128(uint32_t a, uint32_t b)
{
    return __builtin_amdgcn_udot4(b, 0x80402010u, __builtin_amdgcn_udot4(a, 0x08040201u, 0u, false), false);
}
// ... of four dwords -> 16 bits in position order
__device__ __forceinline__ uint32_t rx_pack16(uint32_t a, uint32_t b, uint32_t c, uint32_t d)
{
    return (rx_pack8x128(a, b) >> 7) | (rx_pack8x128(c, d) << 1);
}

// ------------------------------------------------------------------------------------------
// the row index: where every row of the block starts and where it ends
// ------------------------------------------------------------------------------------------
// A row starts at a byte that is no row terminator and follows one (or the block's start): blank lines are no rows
// (getnextrow.cpp:38-43); it ends at the first row terminator behind its start; a newline terminates a row iff an even number of
// backslashes precedes it (getnextrow.cpp:45-53,72-77).  So starts and ends alternate, and the i-th end belongs to the i-th start.
// One streaming read of the block.  The waves take 16 KiB spans in order (a ticket each); a wave keeps the starts and ends of its
// span as 16-bit sets per lane and 1 KiB round (32 registers), counts them, learns the number of rows before its span from the spans
// before it (a decoupled look-back over one word per span: count, then running total -- k_pass1's, see there), and writes
//     rowidx[2 i] = start of row i,  rowidx[2 i + 1] = position of its terminator,  rowidx[2 rows] = hi,
// the number of rows + 1 into st->rows_total, and per span whether it holds a backslash at all (bsmap).
constexpr uint32_t RX_SHIFT = 16, RX_SPAN = 1u << RX_SHIFT, RX_ROUND = 4096, RX_ROUNDS = RX_SPAN / RX_ROUND, RX_AHEAD = 4;
constexpr uint32_t RX_BS_SHIFT = 14;          // the backslash map's granule: 16 KiB

typedef uint32_t rx_u32x4 __attribute__((ext_vector_type(4)));
struct RxChunk { rx_u32x4 v[4]; };            // a lane's 64 consecutive bytes of a round

__device__ __forceinline__ RxChunk rx_load64(const uint8_t *in, uint64_t pos, uint32_t lane, uint64_t n)
{
    // lanes whose bytes would start at or past the end of the allocation re-read the last 16-byte piece that starts inside it;
    // callers mask by position
    const uint64_t last = (n - 1) & ~15ull;
    RxChunk c;
#pragma unroll
    for (uint32_t k = 0; k < 4; ++k) {
        const uint64_t p = pos + 64ull * lane + 16u * k;
        c.v[k] = __builtin_nontemporal_load(reinterpret_cast<const rx_u32x4 *>(in + (p <= last ? p : last)));
    }
    return c;
}
// bits i of a lane whose absolute position lp + i lies in [vlo, vhi)
__device__ __forceinline__ uint64_t rx_valid64(uint64_t lp, uint64_t vlo, uint64_t vhi)
{
    const uint32_t lo_i = vlo > lp ? (vlo - lp < 64 ? (uint32_t)(vlo - lp) : 64u) : 0u;
    const uint32_t hi_i = vhi > lp ? (vhi - lp < 64 ? (uint32_t)(vhi - lp) : 64u) : 0u;
    const uint64_t below_hi = hi_i >= 64u ? ~0ull : ((1ull << hi_i) - 1ull), below_lo = lo_i >= 64u ? ~0ull : ((1ull << lo_i) - 1ull);
    return below_hi & ~below_lo;
}

__global__ __launch_bounds__(64) void k_row_index(const RowIndexArgs a)
{
    const uint32_t lane = lane_id();
    P1Status *const stp = a.st;
    for (;;) {
        uint32_t t = 0;
        if (lane == 0) t = atomicAdd(&stp->next_span, 1u);
        const uint32_t si = lane_value(t, 0);
        if (si >= a.nspans) break;
        // (nothing between taking a span and publishing its count may leave the loop: the spans behind it wait for that count)
        const uint64_t s0 = (a.span0 + si) << RX_SHIFT;
        const uint64_t ss = s0 > a.lo ? s0 : a.lo, se = s0 + RX_SPAN < a.hi ? s0 + RX_SPAN : a.hi;
        uint32_t prev_term = ((ss == a.lo) || uniform32(is_row_terminator(a.in, a.lo, ss - 1) ? 1u : 0u)) ? 1u : 0u;
        const uint32_t open = prev_term ? 0u : 1u;                  // a row that began in an earlier span is open at ss: its end is this span's first
        bool bs_before = s0 > a.lo && uniform32(a.in[s0 - 1] == '\\' ? 1u : 0u) != 0;      // the round before held a backslash (its last byte may be one)
        uint64_t sm[RX_ROUNDS], em[RX_ROUNDS];                       // row starts / row ends among this lane's 64 bytes of every round
        uint32_t cnt = 0, nul = 0;                                   // starts | ends << 16 of this lane
        uint32_t bs_granules = 0;                                    // bit g: the span's g-th 16 KiB holds a backslash
        RxChunk q[RX_AHEAD];
#pragma unroll
        for (uint32_t r = 0; r < RX_AHEAD; ++r) q[r] = rx_load64(a.in, s0 + r * RX_ROUND, lane, a.n);
#pragma unroll
        for (uint32_t r = 0; r < RX_ROUNDS; ++r) {
            const uint64_t pos = s0 + r * RX_ROUND, lp = pos + 64ull * lane;
            const RxChunk c = q[r % RX_AHEAD];
            if (r + RX_AHEAD < RX_ROUNDS) q[r % RX_AHEAD] = rx_load64(a.in, s0 + (r + RX_AHEAD) * RX_ROUND, lane, a.n);
            uint64_t s64 = 0, e64 = 0;
            if (pos < se && pos + RX_ROUND > ss) {      // (rounds of the span that lie before the block's start or behind its end: nothing, and prev_term stays)
                const uint32_t H = 0x80808080u, L1 = 0x01010101u;
                uint32_t t16[4], anybs = 0, anynul = 0;
#pragma unroll
                for (uint32_t k = 0; k < 4; ++k) {
                    const rx_u32x4 v = c.v[k];
                    t16[k] = rx_pack16(~rx_nz7b(v.x ^ 0x0a0a0a0au) & H, ~rx_nz7b(v.y ^ 0x0a0a0a0au) & H, ~rx_nz7b(v.z ^ 0x0a0a0a0au) & H, ~rx_nz7b(v.w ^ 0x0a0a0a0au) & H);
                    const uint32_t b0 = v.x ^ 0x5c5c5c5cu, b1 = v.y ^ 0x5c5c5c5cu, b2 = v.z ^ 0x5c5c5c5cu, b3 = v.w ^ 0x5c5c5c5cu;
                    anybs |= ((b0 - L1) & ~b0) | ((b1 - L1) & ~b1) | ((b2 - L1) & ~b2) | ((b3 - L1) & ~b3);             // & H: exact for "some byte is a backslash"
                    anynul |= ((v.x - L1) & ~v.x) | ((v.y - L1) & ~v.y) | ((v.z - L1) & ~v.z) | ((v.w - L1) & ~v.w);     // & H: exact for "some byte is NUL"
                }
                uint64_t term = (uint64_t)(t16[0] | (t16[1] << 16)) | ((uint64_t)(t16[2] | (t16[3] << 16)) << 32);       // bit i: byte i is a newline
                const bool bs_here = __ballot((anybs & H) != 0u) != 0ull;
                if (bs_here | bs_before) {
                    // backslashes about: a newline terminates a row iff an even number of them precedes it (exact, from the image)
                    uint64_t cc = term;
                    while (cc) {
                        const uint32_t i = (uint32_t)__builtin_ctzll(cc);
                        cc &= cc - 1;
                        if (lp + i > a.lo && lp + i < a.n && (backslashes_before(a.in, a.lo, lp + i) & 1u)) term &= ~(1ull << i);
                    }
                }
                if (bs_here) bs_granules |= 1u << (r * RX_ROUND >> RX_BS_SHIFT);
                bs_before = bs_here;
                uint64_t vm = ~0ull;
                if (pos < ss || pos + RX_ROUND > se) vm = rx_valid64(lp, ss, se);             // the block's first / last round
                if (vm == ~0ull) nul |= anynul & H;
                else if (anynul & H) {      // (bytes outside the block do not count: the exact set)
                    uint64_t z = 0;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        const rx_u32x4 v = c.v[k];
                        z |= (uint64_t)rx_pack16(~rx_nz7b(v.x) & H, ~rx_nz7b(v.y) & H, ~rx_nz7b(v.z) & H, ~rx_nz7b(v.w) & H) << (16u * k);
                    }
                    if (z & vm) nul |= H;
                }
                term &= vm;
                // "the byte before mine is a row terminator"
                uint64_t pt = (term << 1) | (uint64_t)wave_shr1((uint32_t)(term >> 63));
                if (prev_term) {                                     // ... the byte before the span's (or the round's) first
                    const uint64_t first = pos > ss ? pos : ss;
                    if (first >= lp && first < lp + 64) pt |= 1ull << (uint32_t)(first - lp);
                }
                s64 = pt & ~term & vm;
                e64 = term & ~pt;
                prev_term = lane_value((uint32_t)(term >> 63), 63);      // the round's last byte, for the next round
                cnt += (uint32_t)__builtin_popcountll(s64) + ((uint32_t)__builtin_popcountll(e64) << 16);
            }
            sm[r] = s64; em[r] = e64;
        }
        uint32_t tot2;
        (void)wave_excl_sum(cnt, lane, tot2);
        tot2 = uniform32(tot2);
        const uint32_t rows_here = tot2 & 0xffffu;
        if (__ballot(nul != 0)) stp->nul_seen = 1;
        if (lane < (RX_SPAN >> RX_BS_SHIFT)) a.bsmap[(uint64_t)si * (RX_SPAN >> RX_BS_SHIFT) + lane] = (uint8_t)((bs_granules >> lane) & 1u);
        // ---- rows that start in the spans before this one (k_pass1's look-back) ----
        uint64_t before = 0;
        bool give_up = false;
        if (si) {
            __hip_atomic_store(a.span_state + si, (1ull << 62) | rows_here, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int64_t j = (int64_t)si - 1;
            uint32_t spins = 0;
            for (;;) {
                const int64_t idx = j - (int64_t)lane;
                uint64_t w = 2ull << 62;
                if (idx >= 0) w = __hip_atomic_load(a.span_state + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t flag = (uint32_t)(w >> 62);
                const uint64_t known = __ballot(flag == 2u), pending = __ballot(flag == 0u);
                const int first = known ? lowest_bit(known) : 64;
                if (pending & mask_lt(first)) {
                    if (++spins > (1u << 22)) { stp->fold_abort = 1; give_up = true; break; }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                uint32_t lo32 = (lane <= (uint32_t)first) ? (uint32_t)w : 0u, hi32 = (lane <= (uint32_t)first) ? (uint32_t)(w >> 32) & 0x3fffffffu : 0u;
                uint32_t tl, tm, th;
                (void)wave_excl_sum(lo32 & 0xffffffu, lane, tl);
                (void)wave_excl_sum(lo32 >> 24, lane, tm);
                (void)wave_excl_sum(hi32, lane, th);
                before += (uint64_t)uniform32(tl) + ((uint64_t)uniform32(tm) << 24) + ((uint64_t)uniform32(th) << 32);
                if (first < 64) break;
                j -= 64;
            }
        }
        __hip_atomic_store(a.span_state + si, (2ull << 62) | (before + rows_here), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (si + 1 == a.nspans) {
            const uint64_t starts = before + rows_here, ends = before - open + (tot2 >> 16);
            stp->rows_total = starts + 1;
            if (starts < a.rows_cap + 2) a.rowidx[2 * starts] = a.hi;
            // (a block that is looked at through a window may end inside a row: that row has no terminator in view -- `hi` says so)
            if (ends < starts && starts - 1 < a.rows_cap + 2) a.rowidx[2 * (starts - 1) + 1] = a.hi;
        }
        if (give_up) continue;
        // ---- the positions ----
        uint64_t sbase = before, ebase = before - open;              // (open: at least one row precedes)
#pragma unroll
        for (uint32_t r = 0; r < RX_ROUNDS; ++r) {
            uint64_t s64 = sm[r], e64 = em[r];
            if (!__ballot((s64 | e64) != 0ull)) continue;
            const uint32_t c2 = (uint32_t)__builtin_popcountll(s64) | ((uint32_t)__builtin_popcountll(e64) << 16);
            uint32_t t2;
            const uint32_t ex = wave_excl_sum(c2, lane, t2);
            t2 = uniform32(t2);
            const uint64_t lp = s0 + r * RX_ROUND + 64ull * lane;
            uint64_t k = sbase + (ex & 0xffffu);
            while (s64) {
                const uint32_t i = (uint32_t)__builtin_ctzll(s64);
                s64 &= s64 - 1;
                if (k < a.rows_cap + 2) a.rowidx[2 * k] = lp + i;
                ++k;
            }
            k = ebase + (ex >> 16);
            while (e64) {
                const uint32_t i = (uint32_t)__builtin_ctzll(e64);
                e64 &= e64 - 1;
                if (k < a.rows_cap + 2) a.rowidx[2 * k + 1] = lp + i;
                ++k;
            }
            sbase += t2 & 0xffffu; ebase += t2 >> 16;
        }
    }
}

void launch_row_index(const RowIndexArgs &a, hipStream_t s)
{
    if (!a.nspans) return;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    // (a grid of what is resident at once: later workgroups would only take later tickets)
    const uint32_t cap = (uint32_t)cus * 12u;
    hipLaunchKernelGGL(k_row_index, dim3(a.nspans < cap ? a.nspans : cap), dim3(64), 0, s, a);
}

// ------------------------------------------------------------------------------------------
// pass 1, a lane per row
// ------------------------------------------------------------------------------------------
// LDS of one wave (RW_WAVE_LDS bytes; lane l):
//   ring     72 rows of 64 dwords: dword j of the lane's 256-byte ring at (j * 64 + l) * 4; rows 64-71 repeat rows 0-7, so eight
//            consecutive dwords are read from any start without wrapping (ds_read2st64_b32: two rows per instruction)
//   bitmap   4 x 64 u64: bit i of word u = byte i of the ring's u-th 64 bytes is a tab or a newline.  The bytes are classified ONCE,
//            when they arrive (aligned dwords straight from the load's registers, no alignment shuffle), not once per field
//   stage_n / stage_t   records of the row being walked: 32 bytes of numeric ones per lane; text ones in a ring of three 16-byte pieces --
//            a piece is written out once the slots of all its fields have been looked at, RW_DEPTH steps behind the walk
constexpr uint32_t RW_RING_ROWS = 64;                 // dwords of a lane's ring (256 bytes of its row)
constexpr uint32_t RW_MIRROR = 8;
constexpr uint32_t RW_RING_BYTES = (RW_RING_ROWS + RW_MIRROR) * 256u;
constexpr uint32_t RW_UNIT = 64;                      // bytes a lane loads per refill; RW_UNITS of them make the ring
constexpr uint32_t RW_UNITS = RW_RING_ROWS * 4u / RW_UNIT;
constexpr uint32_t RW_BITMAP_BYTES = RW_UNITS * 64u * 8u;
constexpr uint32_t RW_BEHIND = 16;                    // bytes kept behind the cursor (an integer's last 16 characters are read back from its end)
constexpr int32_t RW_NEED = 40;                       // bytes a step may look at ahead of the cursor (32 bits of bitmap, 28 bytes of window from an aligned dword)
constexpr int32_t RW_LAND = 68;                       // a lane with fewer bytes ahead needs the refill that is in flight NOW (a step takes at most 25 + 1 on the fast path)
constexpr int32_t RW_ISSUE = 128;                     // a lane with no more bytes ahead than this starts a refill round ...
constexpr int32_t RW_ROOM = (int32_t)(RW_RING_ROWS * 4u - RW_UNIT - RW_BEHIND);      // ... which every lane with at most this many ahead joins
constexpr uint32_t RW_STAGE_N = 40;                             // staged numeric records per lane: 32 bytes (+ 8: lanes on different banks)
constexpr uint32_t RW_STAGE_T = 56;                             // staged text records per lane: a ring of three 16-byte pieces (+ 8)
constexpr uint32_t RW_WAVE_LDS = RW_RING_BYTES + RW_BITMAP_BYTES + 64u * (RW_STAGE_N + RW_STAGE_T);
constexpr uint32_t RW_LUT_ROWS = 25;
constexpr uint32_t RW_DQ_CHUNK = 256;

struct RwLds { uint32_t waves, lut, meta, mn, mx, seen, total; };
__host__ __device__ inline RwLds rw_layout(uint32_t ncols, uint32_t nnum, uint32_t wpb)
{
    RwLds L;
    L.waves = 0;
    L.lut = wpb * RW_WAVE_LDS;
    L.meta = L.lut + RW_LUT_ROWS * 32u;
    L.mn = (L.meta + 4u * ncols + 7u) & ~7u;
    L.mx = L.mn + 8u * nnum;
    L.seen = L.mx + 8u * nnum;
    L.total = (L.seen + ncols + 15u) & ~15u;
    return L;
}

__device__ __forceinline__ uint64_t rw_u64(uint32_t lo, uint32_t hi) { return ((uint64_t)hi << 32) | lo; }
__device__ __forceinline__ uint64_t rw_lds_u64(uint32_t off) { return *reinterpret_cast<const uint64_t *>(zdw_lds + off); }

// the first tab or newline at or behind p, below hi (hi itself if none): rows without a backslash -- every such byte is a boundary
__device__ inline uint64_t rw_scan_plain(const uint8_t *in, uint64_t p, uint64_t hi)
{
    while (p < hi) {
        const uint32_t b = in[p];
        if (b == '\t' || b == '\n') return p;
        ++p;
    }
    return hi;
}
// ... with escapes: a tab or newline is a boundary iff an even number of backslashes precedes it inside the field (Q1, Q4; a field
// begins behind a boundary, so the run cannot reach further back)
__device__ inline uint64_t rw_scan_careful(const uint8_t *in, uint64_t p, uint64_t hi)
{
    uint32_t run = 0;
    while (p < hi) {
        const uint32_t b = in[p];
        if ((b == '\t' || b == '\n') && !(run & 1u)) return p;
        run = b == '\\' ? run + 1u : 0u;
        ++p;
    }
    return hi;
}

typedef uint32_t rw_u32x4 __attribute__((ext_vector_type(4)));

// Arguments needed on rare paths or once per group of rows are read from the kernel-argument segment where they are used instead of
// living in scalar registers for the whole kernel (k_pass1.hip explains: the kernel holds more wave-uniform values than there are SGPRs)
template <typename T>
__device__ __forceinline__ T rw_kernarg_at(uint32_t off)
{
    typedef __attribute__((address_space(4))) const uint8_t *kptr;
    kptr ka = (kptr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    return *reinterpret_cast<__attribute__((address_space(4))) const T *>(ka + off);
}
#define RW_COLD(field) rw_kernarg_at<decltype(Pass1Args::field)>((uint32_t)offsetof(Pass1Args, field))

// ---- loads in flight across steps -------------------------------------------------------------------------------------------
// A text field's table slot is requested in one step and looked at RW_DEPTH steps later; a lane's next 64 bytes are requested when it
// has 128 left and stored into its ring when it has 68.  Between request and use such data must not be a C++ value: the compiler,
// which places its own s_waitcnt in front of the first use of a load's result, would wait at once wherever it copies the value (the
// loop over the columns is unrolled RW_DEPTH times: copies at every join), and with the load hidden in an asm statement it copies
// registers whose data has not arrived (both seen: the first versions of this kernel).  So the loads go into ACCUMULATION registers
// named in the asm text (a[0:47]; the register allocator never touches them -- the kernel has no MFMA and does not spill), and
// a second asm statement waits and moves them into ordinary registers (v_accvgpr_read).  Loads return in the order they were issued,
// so "the loads I want are done" is s_waitcnt vmcnt(number of loads issued behind them), counted in a wave-uniform sequence number
// (stores and the rare other loads in between only make the wait stricter).
__device__ __forceinline__ void rw_wait_loads(uint32_t behind)
{
#define RW_W(n) asm volatile("s_waitcnt vmcnt(" #n ")" : : : "memory")
    switch (behind >> 1) {
    case 0: RW_W(0); break; case 1: RW_W(2); break; case 2: RW_W(4); break; case 3: RW_W(6); break;
    case 4: RW_W(8); break; case 5: RW_W(10); break; case 6: RW_W(12); break; case 7: RW_W(14); break;
    case 8: RW_W(16); break; case 9: RW_W(18); break;
    default: RW_W(20); break;
    }
#undef RW_W
}
template <int J> __device__ __forceinline__ void rw_slot_request(const uint64_t *sp);
template <int J> __device__ __forceinline__ void rw_slot_take(rw_u32x4 &lo, rw_u32x4 &hi);
#define RW_SLOT_FNS(J, A0, A1, A2, A3, A4, A5, A6, A7) \
    template <> __device__ __forceinline__ void rw_slot_request<J>(const uint64_t *sp) \
    { \
        asm volatile("global_load_dwordx4 a[" #A0 ":" #A3 "], %0, off\n\tglobal_load_dwordx4 a[" #A4 ":" #A7 "], %0, off offset:16" \
                     : : "v"(sp) : "memory", "a" #A0, "a" #A1, "a" #A2, "a" #A3, "a" #A4, "a" #A5, "a" #A6, "a" #A7); \
    } \
    template <> __device__ __forceinline__ void rw_slot_take<J>(rw_u32x4 &lo, rw_u32x4 &hi) \
    { \
        asm volatile("v_accvgpr_read_b32 %0, a" #A0 "\n\tv_accvgpr_read_b32 %1, a" #A1 "\n\tv_accvgpr_read_b32 %2, a" #A2 "\n\tv_accvgpr_read_b32 %3, a" #A3 "\n\t" \
                     "v_accvgpr_read_b32 %4, a" #A4 "\n\tv_accvgpr_read_b32 %5, a" #A5 "\n\tv_accvgpr_read_b32 %6, a" #A6 "\n\tv_accvgpr_read_b32 %7, a" #A7 \
                     : "=v"(lo.x), "=v"(lo.y), "=v"(lo.z), "=v"(lo.w), "=v"(hi.x), "=v"(hi.y), "=v"(hi.z), "=v"(hi.w) : : "memory"); \
    }
RW_SLOT_FNS(0, 0, 1, 2, 3, 4, 5, 6, 7)
RW_SLOT_FNS(1, 8, 9, 10, 11, 12, 13, 14, 15)
RW_SLOT_FNS(2, 16, 17, 18, 19, 20, 21, 22, 23)
RW_SLOT_FNS(3, 24, 25, 26, 27, 28, 29, 30, 31)
#undef RW_SLOT_FNS
// a lane's next 64 bytes: a[32:47]
__device__ __forceinline__ void rw_unit_request(uint32_t off, const uint8_t *base)
{
    asm volatile("global_load_dwordx4 a[32:35], %0, %1\n\t"
                 "global_load_dwordx4 a[36:39], %0, %1 offset:16\n\t"
                 "global_load_dwordx4 a[40:43], %0, %1 offset:32\n\t"
                 "global_load_dwordx4 a[44:47], %0, %1 offset:48"
                 : : "v"(off), "s"(base) : "memory", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
}
__device__ __forceinline__ void rw_unit_take(rw_u32x4 (&v)[4])
{
    asm volatile("v_accvgpr_read_b32 %0, a32\n\tv_accvgpr_read_b32 %1, a33\n\tv_accvgpr_read_b32 %2, a34\n\tv_accvgpr_read_b32 %3, a35\n\t"
                 "v_accvgpr_read_b32 %4, a36\n\tv_accvgpr_read_b32 %5, a37\n\tv_accvgpr_read_b32 %6, a38\n\tv_accvgpr_read_b32 %7, a39"
                 : "=v"(v[0].x), "=v"(v[0].y), "=v"(v[0].z), "=v"(v[0].w), "=v"(v[1].x), "=v"(v[1].y), "=v"(v[1].z), "=v"(v[1].w) : : "memory");
    asm volatile("v_accvgpr_read_b32 %0, a40\n\tv_accvgpr_read_b32 %1, a41\n\tv_accvgpr_read_b32 %2, a42\n\tv_accvgpr_read_b32 %3, a43\n\t"
                 "v_accvgpr_read_b32 %4, a44\n\tv_accvgpr_read_b32 %5, a45\n\tv_accvgpr_read_b32 %6, a46\n\tv_accvgpr_read_b32 %7, a47"
                 : "=v"(v[2].x), "=v"(v[2].y), "=v"(v[2].z), "=v"(v[2].w), "=v"(v[3].x), "=v"(v[3].y), "=v"(v[3].z), "=v"(v[3].w) : : "memory");
}

#ifdef ZDW_EXPERIMENTS
#define RW_TICK(i) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = (unsigned long long)__builtin_readcyclecounter(); \
        asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); dbg_t##i += now_ - dbg_last; dbg_last = now_; } while (0)
#else
#define RW_TICK(i) do { } while (0)
#endif
constexpr uint32_t RW_DEPTH = 4;                      // steps between a slot's request and the look at it
struct RwProbe {                                      // per lane (the slot's 32 bytes travel in a[8 J : 8 J + 7])
    uint64_t c0, c1, c2, slot;                        // the string's first 24 bytes (masked), the slot's index
    uint32_t len, tag, fs;                            // its length, hash tag, start (relative to the group's base)
};
struct RwProbeU { uint64_t wantm; uint32_t col, stage, mark; };      // wave-uniform: lanes with a request, its column, where its record is staged, sequence number

template <bool TRACK, bool FIND>
__global__ __launch_bounds__(64 * RW_WPB_MAX, 1) void k_pass1_rows(const Pass1Args a, const RowsArgs ra)
{
    const uint32_t wv = uniform32(threadIdx.x >> 6);
    const uint32_t wpb = blockDim.x >> 6;
    const uint32_t lane = lane_id();
    const uint32_t ncols = a.ncols;
    const RwLds L = rw_layout(ncols, a.nnum, wpb);
    ColTables t;
    t.meta = L.meta; t.mn = L.mn; t.mx = L.mx; t.seen = L.seen; t.cols_in_lds = true;
    for (uint32_t i = threadIdx.x; i < RW_LUT_ROWS * 8u; i += blockDim.x) {
        const uint32_t row = i >> 3, b0 = (i & 7u) * 4u;
        const uint32_t nb = row > b0 ? (row - b0 < 4u ? row - b0 : 4u) : 0u;
        *reinterpret_cast<uint32_t *>(zdw_lds + L.lut + 4u * i) = nb >= 4u ? 0xffffffffu : ((1u << (8u * nb)) - 1u);
    }
    for (uint32_t c = threadIdx.x; c < ncols; c += blockDim.x) {
        *reinterpret_cast<uint32_t *>(zdw_lds + t.meta + 4 * c) = pack_col_meta(RW_COLD(col_kind)[c], (uint32_t)RW_COLD(col_numidx)[c] & 0x3fffu, RW_COLD(col_recoff)[c]);
        zdw_lds[t.seen + c] = 0;
    }
    for (uint32_t i = threadIdx.x; i < a.nnum; i += blockDim.x) {
        reinterpret_cast<uint64_t *>(zdw_lds + t.mn)[i] = ~0ull;
        reinterpret_cast<uint64_t *>(zdw_lds + t.mx)[i] = 0;
    }
    __syncthreads();

    const uint32_t ring = L.waves + wv * RW_WAVE_LDS;                 // this wave's ring
    const uint32_t bitmap = ring + RW_RING_BYTES, stage_n = bitmap + RW_BITMAP_BYTES, stage_t = stage_n + 64u * RW_STAGE_N;
    const uint32_t ring_l = ring + 4u * lane, bitmap_l = bitmap + 8u * lane;
    uint64_t nstore, nwalk;
    {
        P1Status *const stp = RW_COLD(st);
        const uint64_t rows_found = stp->rows_total ? stp->rows_total - 1 : 0;               // (the row index has run: same stream)
        // rows that leave records, and the row behind them: the one at a block's end inside a row (Q9) -- its first fields still count,
        // and the reference has read it (m_LongestLine)
        nstore = rows_found < a.row_limit ? rows_found : a.row_limit;
        nwalk = nstore < rows_found ? nstore + 1 : nstore;
    }
    const uint32_t K = ra.rows_per_lane;
    IdAlloc ids;
    ids.cur = ids.end = ids.taken = 0;
    uint32_t dq_cur = 0, dq_end = 0;                                  // this wave's chunk of deferred-queue entries
    uint32_t new_entries = 0;
    uint32_t vm_seq = 0;                                              // loads of the kind above issued so far
#ifdef ZDW_EXPERIMENTS
    unsigned long long dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0, dbg_t3 = 0, dbg_t4 = 0, dbg_t5 = 0, dbg_t6 = 0, dbg_t7 = 0, dbg_t8 = 0, dbg_t9 = 0, dbg_t10 = 0, dbg_t11 = 0,
                       dbg_last = (unsigned long long)__builtin_readcyclecounter();
#endif
    RwProbe pq[RW_DEPTH];
    RwProbeU pu[RW_DEPTH];
#pragma unroll
    for (uint32_t j = 0; j < RW_DEPTH; ++j) {
        pq[j].c0 = pq[j].c1 = pq[j].c2 = pq[j].slot = 0; pq[j].len = pq[j].tag = pq[j].fs = 0;
        pu[j].wantm = 0; pu[j].col = pu[j].stage = pu[j].mark = 0;
    }

    for (;;) {
        {
            P1Status *const stp = RW_COLD(st);
            if (uniform32(*reinterpret_cast<volatile const uint32_t *>(&stp->overflow))) break;
            if (new_entries >= 128) {
                uint32_t seen_total = 0;
                if (lane == 0) seen_total = atomicAdd(&stp->entries, new_entries) + new_entries;
                seen_total = lane_value(seen_total, 0);
                new_entries = 0;
                if (seen_total >= RW_COLD(max_entries)) { stp->overflow = 1; break; }
            }
        }
        uint32_t tk = 0;
        if (lane == 0) tk = atomicAdd(&RW_COLD(st)->next_group, 1u);
        const uint64_t g0 = (uint64_t)lane_value(tk, 0) * 64u * K;   // the group's first row
        if (g0 >= nwalk) break;
        const uint64_t my0 = g0 + (uint64_t)lane * K;
        // positions are kept relative to the group's base (a multiple of 64 below its first row): 32-bit offsets from one scalar pointer
        const uint64_t gbase = uniform64(ra.rowidx[2 * g0]) & ~63ull;
        const uint8_t *const imgb = RW_COLD(in) + gbase;
        // (a piece past the end of the image: its last piece again -- bytes nobody looks at)
        uint32_t last16;
        {
            const uint64_t l = ((RW_COLD(n) - 1) & ~15ull) - gbase;
            last16 = l < 0xfffffff0ull ? (uint32_t)l : 0xfffffff0u;
        }
        uint32_t cursor = 0, fill = 0;                                // relative positions: next byte to look at; end of what the ring holds
        int32_t avail = 0;                                            // fill - cursor
        uint64_t okm = 0;                                             // lanes whose ring holds what their cursor needs
        uint64_t pendm = 0;
        uint32_t pend_mark = 0;

        // a unit of 64 bytes (four 16-byte pieces in v) into the ring rows of relative position f (a multiple of 64), and its boundaries
        auto store_unit = [&](uint32_t f, const rw_u32x4 (&v)[4]) {
            const uint32_t r0 = (f >> 2) & (RW_RING_ROWS - 1u);
            uint32_t *q = reinterpret_cast<uint32_t *>(zdw_lds + ring_l + r0 * 256u);
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) { q[(4 * k) * 64] = v[k].x; q[(4 * k + 1) * 64] = v[k].y; q[(4 * k + 2) * 64] = v[k].z; q[(4 * k + 3) * 64] = v[k].w; }
            if (r0 == 0) {      // the ring's first rows once more behind its last
                uint32_t *m = reinterpret_cast<uint32_t *>(zdw_lds + ring_l + RW_RING_ROWS * 256u);
                m[0] = v[0].x; m[64] = v[0].y; m[128] = v[0].z; m[192] = v[0].w; m[256] = v[1].x; m[320] = v[1].y; m[384] = v[1].z; m[448] = v[1].w;
            }
            uint32_t b16[4];
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) b16[k] = rx_pack16(rx_tab_or_nl7(v[k].x), rx_tab_or_nl7(v[k].y), rx_tab_or_nl7(v[k].z), rx_tab_or_nl7(v[k].w));
            uint32_t *bm = reinterpret_cast<uint32_t *>(zdw_lds + bitmap_l + ((f >> 6) & (RW_UNITS - 1u)) * 512u);
            bm[0] = b16[0] | (b16[1] << 16); bm[1] = b16[2] | (b16[3] << 16);
        };
        // ... loaded the ordinary way (the compiler waits where it is used): what starts a group, follows a field of the general form, or
        // reaches the image's end
        auto load_unit = [&](uint32_t f, rw_u32x4 (&v)[4]) {
#pragma unroll
            for (uint32_t k = 0; k < 4; ++k) {
                const uint32_t o = f + 16u * k;
                v[k] = *reinterpret_cast<const rw_u32x4 *>(imgb + (o <= last16 ? o : last16));
            }
        };
        // the refill in flight lands in the ring
        auto land = [&](bool arrived) __attribute__((always_inline)) {
            if (!arrived) rw_wait_loads(vm_seq - pend_mark);
            rw_u32x4 pend[4];
            rw_unit_take(pend);
            if (lane_in(pendm)) { store_unit(fill, pend); fill += RW_UNIT; avail += (int32_t)RW_UNIT; }
            pendm = 0;
        };
        // lanes (mask m) whose ring does not hold what their cursor needs: three units from 16 bytes behind the cursor on, waited for
        auto resync = [&](uint64_t m) __attribute__((always_inline)) {
            if (pendm) land(false);
            if (lane_in(m)) {
                const uint32_t f0 = cursor >= RW_BEHIND ? (cursor - RW_BEHIND) & ~63u : 0u;       // (bytes before the group's base are never looked at)
#pragma unroll
                for (uint32_t u = 0; u < 3; ++u) {
                    rw_u32x4 v[4];
                    load_unit(f0 + u * RW_UNIT, v);
                    store_unit(f0 + u * RW_UNIT, v);
                }
                fill = f0 + 3u * RW_UNIT;
                avail = (int32_t)(fill - cursor);
            }
            okm |= m;
        };

        for (uint32_t ri = 0; ri < K; ++ri) {
            const uint64_t row = my0 + ri;
            const bool act = row < nwalk;
            const uint64_t actm = __ballot(act);
            if (!actm) break;
            uint64_t S = 0, T = 0, NX = 0;
            if (act) { S = ra.rowidx[2 * row]; T = ra.rowidx[2 * row + 1]; NX = ra.rowidx[2 * row + 2]; }
            // (what the index says is checked before anything is addressed by it: a row lies inside the block, its successor behind it)
            const bool sane = act && S >= RW_COLD(lo) && S < RW_COLD(hi) && NX > S && NX <= RW_COLD(hi);
            if (act && !sane) { S = RW_COLD(lo); T = S; NX = S + 1; }
            // does a backslash lie anywhere in [S, NX)?  (the row index says so per 16 KiB)
            bool careful = false;
            if (act) {
                const uint64_t sp0 = (S >> RX_BS_SHIFT) - ra.bs_span0, sp1 = ((NX - 1) >> RX_BS_SHIFT) - ra.bs_span0;
                if (sp1 - sp0 > 3) careful = true;
                else for (uint64_t sp = sp0; sp <= sp1; ++sp) careful |= ra.bsmap[sp] != 0;
            }
            const uint64_t carefulm = __ballot(careful);
            const uint32_t my_cols = act ? (row < nstore ? ncols : a.partial_fields) : 0u;         // fields of this row that count
            const uint64_t fullm = __ballot(my_cols == ncols);        // (all but a block's last, partial row)
            uint64_t deadm = ~actm | __ballot(act && !sane);            // lanes that take no part in the rest of this row
            cursor = (uint32_t)(S - gbase);
            avail = (int32_t)(fill - cursor);
            {
                okm &= ~carefulm;
                const uint64_t needm = actm & ~carefulm & (~okm | BAL(avail < RW_NEED) | BAL(avail > (int32_t)(RW_RING_ROWS * 4u - RW_BEHIND)));
                if (needm) resync(needm);
            }
            // staged records of the row: the region's first record offset and the bytes staged (wave-uniform).  Both regions of a row's
            // records are laid out in column order (zdw_capi.hip: numeric columns, then text columns), so what is staged is contiguous.
            uint32_t off_n = 0, nb_n = 0;
            uint32_t t_base = 0, t_cnt = 0, t_flushed = 0, t_cnt_prev = 0;      // text records: offset of the row's first; staged so far; written out; staged when the group before began

            // records staged for the 64 rows of this step go out as whole pieces, neighbouring lanes on neighbouring pieces of a row
            // rows of the group at this row iteration that leave records: row (g0 + r K + ri) for r < rmax
            uint32_t rmax;
            {
                const uint64_t first = g0 + ri;
                rmax = first >= nstore ? 0u : (uint32_t)((nstore - first + K - 1) / K < 64u ? (nstore - first + K - 1) / K : 64u);
            }
            uint8_t *const rec_ri = a.rec + (g0 + ri) * a.rec_stride;          // record row of lane 0's row; lane r's: + r K stride
            const uint32_t kstride = K * a.rec_stride;
            // nb staged bytes (a multiple of 4) at byte `src` of every lane's staging row (stride `stride`) go to byte `off` of the rows'
            // records: 16-byte pieces first, neighbouring lanes on neighbouring pieces of a row, then single dwords
            auto flush = [&](uint32_t stage, uint32_t stride, uint32_t src, uint32_t off, uint32_t nb) __attribute__((always_inline)) {
                if (!nb) return;
                __builtin_amdgcn_wave_barrier();
                const uint32_t n16 = nb >> 4, rem = (nb & 15u) >> 2;
                for (uint32_t e0 = 0; e0 < 64u * n16; e0 += 64u) {
                    const uint32_t e = e0 + lane;
                    const uint32_t r = n16 == 1 ? e : e >> 1, k = e - r * n16;          // (at most two whole pieces at a time)
                    const uint64_t *q = reinterpret_cast<const uint64_t *>(zdw_lds + stage + r * stride + src + 16u * k);
                    const uint64_t v0 = q[0], v1 = q[1];
                    if (r < rmax) {
                        // (record offsets are multiples of 4 only; a global store takes any alignment)
                        typedef uint32_t u32x4a4 __attribute__((ext_vector_type(4), aligned(4)));
                        u32x4a4 x; x.x = (uint32_t)v0; x.y = (uint32_t)(v0 >> 32); x.z = (uint32_t)v1; x.w = (uint32_t)(v1 >> 32);
                        *reinterpret_cast<u32x4a4 *>(rec_ri + (__umul24(r, kstride) + off + 16u * k)) = x;
                    }
                }
                for (uint32_t e0 = 0; e0 < 64u * rem; e0 += 64u) {
                    const uint32_t e = e0 + lane;
                    const uint32_t r = rem == 1 ? e : (rem == 2 ? e >> 1 : e / 3u), k = e - r * rem;
                    const uint32_t v = *reinterpret_cast<const uint32_t *>(zdw_lds + stage + r * stride + src + 16u * n16 + 4u * k);
                    if (r < rmax) *reinterpret_cast<uint32_t *>(rec_ri + (__umul24(r, kstride) + off + 16u * n16 + 4u * k)) = v;
                }
                __builtin_amdgcn_wave_barrier();
            };
            // whole pieces of text records whose fields have all found their slots (those staged before the group before began)
            auto flush_text = [&](uint32_t resolved) __attribute__((always_inline)) {
                while (t_flushed + 4u <= resolved) {
                    flush(stage_t, RW_STAGE_T, 16u * ((t_flushed >> 2) % 3u), t_base + 4u * t_flushed, 16u);
                    t_flushed += 4u;
                }
            };
            // fields for the general form: queued for k_pass1_deferred (position | column << 40, length | row << 32)
            auto defer = [&](uint64_t sm, uint64_t abs_start, uint32_t raw_len, uint32_t col) __attribute__((always_inline)) {
                if (!sm) return;
                P1Status *const stp = RW_COLD(st);
                uint64_t *const dq = RW_COLD(dq);
                const uint32_t dq_cap = RW_COLD(dq_cap);
                const uint32_t k = popc64(sm);
                if (dq_end - dq_cur < k) {
                    const uint32_t left = dq_end - dq_cur;
                    if (lane < left && dq_cur + lane < dq_cap) dq[2ull * (dq_cur + lane) + 1] = ~0ull;
                    uint32_t fresh = 0;
                    if (lane == 0) fresh = atomicAdd(&stp->deferred, RW_DQ_CHUNK);
                    dq_cur = lane_value(fresh, 0);
                    dq_end = dq_cur + RW_DQ_CHUNK;
                }
                const uint32_t qb = dq_cur + mbcnt64(sm);
                dq_cur += k;
                if (lane_in(sm)) {
                    if (qb < dq_cap) {
                        dq[2ull * qb] = abs_start | ((uint64_t)col << 40);
                        dq[2ull * qb + 1] = (uint64_t)raw_len | ((uint64_t)row << 32);
                    } else {
                        stp->deferred_lost = 1;
                    }
                }
            };
            // ---- a text field whose slot was requested RW_DEPTH steps ago: the slot decides ----
            auto resolve = [&](auto jc) __attribute__((always_inline)) {
                constexpr int J = decltype(jc)::value;
                RwProbe &P = pq[J];
                RwProbeU &U = pu[J];
                if (!U.wantm) return;
                rw_wait_loads(vm_seq - U.mark);
                rw_u32x4 s_lo, s_hi;
                rw_slot_take<J>(s_lo, s_hi);
                // (a refill requested before this slot has arrived with it -- loads return in order: it costs no wait of its own)
                if (pendm && (int32_t)(U.mark - pend_mark) > 0) land(true);
                SlotWords sw;
                sw.key = ((uint64_t)s_lo.y << 32) | s_lo.x; sw.s2 = ((uint64_t)s_lo.w << 32) | s_lo.z;
                sw.s0 = ((uint64_t)s_hi.y << 32) | s_hi.x; sw.s1 = ((uint64_t)s_hi.w << 32) | s_hi.z;
                uint64_t pm = U.wantm, slowm = 0, newm = 0;
                uint32_t t_id = 0;
                uint64_t *const slots = a.slots;
                for (uint32_t probe = 0; probe < (FIND ? 1u : 4u); ++probe) {
                    if (!pm) break;
                    uint64_t *slot = slots + 4 * P.slot;
                    if (FIND) {
                        // (nobody inserts while this kernel runs: what the ordinary load showed is what the table holds)
                    } else if (probe == 0) {
                        // that was an ordinary load, possibly from this CU's L1: only a complete slot can be taken at its word (zdw_dict_dev.h, request_slot)
                        const uint64_t againm = pm & (BAL(sw.key == 0) | BAL(slot_incomplete(sw)));
                        if (againm) { if (lane_in(againm)) sw = load_slot(slot); }
                    } else if (lane_in(pm)) sw = load_slot(slot);
                    const uint64_t samem = pm & BAL(sw.s0 == P.c0) & BAL(sw.s1 == P.c1) & BAL(sw.s2 == P.c2) & BAL((((uint32_t)(sw.key >> KEY_LEN_SHIFT)) & 0xffu) == P.len);
                    if (lane_in(samem)) {
                        t_id = (uint32_t)sw.key;
                        if (TRACK) note_occurrence(RW_COLD(uniq_pos), t_id - 1u, gbase + P.fs);
                    }
                    pm &= ~samem;
                    if (FIND) break;                                    // everything else: the general form, behind this kernel
                    const uint64_t freem = pm & BAL(sw.key == 0);       // an empty slot: the string is new
                    if (freem) {
                        P1Status *const stp = RW_COLD(st);
                        const uint32_t nid = alloc_ids(ids, freem, stp, ID_CHUNK);
                        uint32_t won = 0;
                        if (lane_in(freem)) {
                            if (nid >= RW_COLD(id_cap)) { stp->overflow = 1; }
                            else {
                                const uint64_t old = claim_slot<false, false>(slot, make_slot_key(P.tag, P.len, nid), nid, gbase + P.fs, P.len, RW_COLD(uniq_pos), RW_COLD(uniq_len), P.c0, P.c1, P.c2, TRACK, RW_COLD(uniq_key));
                                if (old == 0) { won = 1u; t_id = nid + 1; } else won = 2u;
                            }
                        }
                        newm |= BAL(won == 1u);
                        slowm |= BAL(won == 2u);                        // lost the race: general form
                        pm &= ~freem;
                    }
                    const uint64_t halfm = pm & BAL(slot_incomplete(sw));  // claimed, inline copy not complete yet: general form
                    slowm |= halfm; pm &= ~halfm;
                    if (lane_in(pm)) P.slot = (P.slot + 1) & a.slot_mask;
                }
                slowm |= pm;
                if (lane_in(U.wantm & ~slowm)) *reinterpret_cast<uint32_t *>(zdw_lds + stage_t + lane * RW_STAGE_T + U.stage) = t_id;
                new_entries += popc64(newm);
                defer(slowm, gbase + P.fs, P.len, U.col);
                U.wantm = 0;
            };
            auto drain = [&]() __attribute__((always_inline)) {
                resolve(std::integral_constant<int, 0>{}); resolve(std::integral_constant<int, 1>{});
                resolve(std::integral_constant<int, 2>{}); resolve(std::integral_constant<int, 3>{});
            };

            for (uint32_t col0 = 0; col0 < ncols; col0 += RW_DEPTH) {
                // ---- staged records: written out when the next RW_DEPTH columns' records may not fit behind what a region holds
                // ---- (text: every field that has a record there has found its slot by then) ----
                if (nb_n) { flush(stage_n, RW_STAGE_N, 0u, off_n, nb_n); nb_n = 0; }
                flush_text(t_cnt_prev);
                t_cnt_prev = t_cnt;
                auto step = [&](auto jc, const uint32_t col) __attribute__((always_inline)) {
                    constexpr int J = decltype(jc)::value;
                    RwProbe &P = pq[J];
                    RwProbeU &U = pu[J];
                    const uint32_t meta = uniform32(*reinterpret_cast<const uint32_t *>(zdw_lds + t.meta + 4u * col));
                    const uint32_t kind = meta & 3u, recoff = meta >> 16;
                    const uint64_t livem = ~deadm & (fullm | BAL(col < my_cols));
                    const uint64_t fastm = livem & ~carefulm;
                    // ---- where my field ends: the first boundary at or behind the cursor, from the bitmap (32 bits of it) ----
                    const uint32_t fs = cursor;
                    uint32_t len;
                    {
                        const uint32_t u = (fs >> 6) & (RW_UNITS - 1u), o = fs & 63u;
                        const uint32_t *b0 = reinterpret_cast<const uint32_t *>(zdw_lds + bitmap_l + u * 512u);
                        const uint32_t *b1 = reinterpret_cast<const uint32_t *>(zdw_lds + bitmap_l + ((u + 1u) & (RW_UNITS - 1u)) * 512u);
                        const uint32_t w0 = b0[0], w1 = b0[1], w2 = b1[0];
                        const bool upper = o >= 32u;
                        const uint32_t lo = upper ? w1 : w0, hi = upper ? w2 : w1;
                        const uint32_t bits = __builtin_amdgcn_alignbit(hi, lo, o);      // (the shift is o mod 32)
                        len = bits ? (uint32_t)__builtin_ctz(bits) : 32u;
                    }
                    RW_TICK(0);          // 0: loop overhead, column facts, the field's end from the bitmap
                    const uint32_t limit = kind == K_TEXT ? 24u : 16u;
                    // fields the fast path does not take (the general form takes them): longer than it reads in one piece -- and, where the
                    // bitmap's 32 bits show no boundary or the row holds a backslash, their end from the image, byte by byte
                    const uint64_t scanm = (livem & carefulm) | (fastm & BAL(len >= 32u));
                    uint64_t slowm = scanm | (fastm & BAL(len > limit));
                    if (scanm) {
                        if (lane_in(scanm)) {
                            const uint64_t p0 = gbase + fs, hi = RW_COLD(hi);
                            const uint64_t e = careful ? rw_scan_careful(RW_COLD(in), p0, hi) : rw_scan_plain(RW_COLD(in), p0, hi);
                            uint32_t kill = e >= hi ? 1u : 0u;              // the row runs into the block's end: not a row of ncols fields
                            const uint64_t l64 = e - p0;
                            if (l64 >= (1ull << 30)) { RW_COLD(st)->nul_seen = 2; kill = 1u; }
                            len = (uint32_t)l64;
                            if (kill) len = 0x40000000u;                    // (marks the lane dead below)
                        }
                        deadm |= scanm & BAL(len == 0x40000000u);
                        slowm &= ~deadm;                                    // (a lane that ran into the block's end has no field to queue)
                    }
                    cursor = fs + len + 1u;
                    avail -= (int32_t)(len + 1u);
                    if (scanm) okm &= ~(scanm & BAL(avail < RW_NEED));       // (the cursor may have left what the ring holds)
                    RW_TICK(1);          // 1: general-form fields of this step
                    // ---- the text field of RW_DEPTH steps ago ----
                    resolve(jc);
                    RW_TICK(2);          // 2: the look at the slot of RW_DEPTH steps ago
                    // ---- a refill that is needed now: wait for it ----
                    if (pendm) {
                        if (fastm & okm & BAL(avail < RW_LAND)) land(false);
                    }
                    RW_TICK(3);          // 3: a refill that had to land now
                    if (kind == K_TEXT) {
                        // ---- text-like column, <= 24 bytes: one table slot decides; requested here, looked at RW_DEPTH steps on ----
                        const uint64_t wantm = fastm & ~slowm & BAL(len != 0u);
                        if (!t_cnt) t_base = recoff;
                        const uint32_t so = 4u * (t_cnt % 12u);             // (text records of a row lie in column order: record t_cnt is at t_base + 4 t_cnt)
                        // the record until then (and for good where the field is empty or goes to the general form): 0
                        *reinterpret_cast<uint32_t *>(zdw_lds + stage_t + lane * RW_STAGE_T + so) = 0u;
                        ++t_cnt;
                        if (wantm) {
                            if (lane == 0) zdw_lds[t.seen + col] = 1;
                            if (lane_in(wantm)) {
                                const uint32_t wrow = (fs >> 2) & (RW_RING_ROWS - 1u), sh = fs & 3u;
                                const uint32_t *wq = reinterpret_cast<const uint32_t *>(zdw_lds + ring_l + wrow * 256u);
                                uint32_t d[7], w[6];
#pragma unroll
                                for (uint32_t k = 0; k < 7; ++k) d[k] = wq[64 * k];
#pragma unroll
                                for (uint32_t k = 0; k < 6; ++k) w[k] = __builtin_amdgcn_alignbyte(d[k + 1], d[k], sh);
                                const uint32_t lo = L.lut + 32u * len;
                                P.c0 = rw_u64(w[0], w[1]) & rw_lds_u64(lo); P.c1 = rw_u64(w[2], w[3]) & rw_lds_u64(lo + 8); P.c2 = rw_u64(w[4], w[5]) & rw_lds_u64(lo + 16);
                                const uint64_t acc = nh24(P.c0, P.c1, P.c2);
                                P.tag = hash_tag(acc);
                                P.slot = hash_place(acc) & a.slot_mask;
                                P.len = len; P.fs = fs;
                                // an ORDINARY load (zdw_dict_dev.h, request_slot: a slot that reads complete is right however stale; anything
                                // else is read again at agent scope)
                                rw_slot_request<J>(a.slots + 4 * P.slot);
                            }
                            vm_seq += 2u;
                            U.wantm = wantm; U.col = col; U.stage = so; U.mark = vm_seq;
                        }
                    } else {
                        const uint64_t wantm = fastm & ~slowm & BAL(len != 0u);
                        uint64_t n_rec = 0;
                        if (wantm) {
                            uint32_t junk = 0;
                            if (lane_in(wantm)) {
                                uint64_t v_rec = 0, v_range = 0;
                                if (kind == K_INT) {
                                    // the field's last 16 characters, read back from its end (what precedes the field is masked)
                                    const uint32_t k = len > 8u ? len - 8u : 0u;
                                    const uint64_t m2 = rw_lds_u64(L.lut + 32u * (8u + k - len)), m1 = rw_lds_u64(L.lut + 32u * (8u - k));
                                    const uint64_t Z = 0x3030303030303030ull;
                                    const uint32_t bp = fs + len - 16u;
                                    const uint32_t brow = (bp >> 2) & (RW_RING_ROWS - 1u), bsh = bp & 3u;
                                    const uint32_t *bq = reinterpret_cast<const uint32_t *>(zdw_lds + ring_l + brow * 256u);
                                    uint32_t bd[5], nw[4];
#pragma unroll
                                    for (uint32_t k2 = 0; k2 < 5; ++k2) bd[k2] = bq[64 * k2];
#pragma unroll
                                    for (uint32_t k2 = 0; k2 < 4; ++k2) nw[k2] = __builtin_amdgcn_alignbyte(bd[k2 + 1], bd[k2], bsh);
                                    const uint64_t w2 = (rw_u64(nw[2], nw[3]) & ~m2) | (Z & m2);
                                    const uint64_t w1 = (rw_u64(nw[0], nw[1]) & ~m1) | (Z & m1);
                                    const uint32_t r0 = (uint32_t)w1, r1 = (uint32_t)(w1 >> 32), r2 = (uint32_t)w2, r3 = (uint32_t)(w2 >> 32);
                                    const uint32_t t0 = r0 - 0x30303030u, t1 = r1 - 0x30303030u, t2 = r2 - 0x30303030u, t3 = r3 - 0x30303030u;
                                    junk = nondigit_bits(r0, t0) | nondigit_bits(r1, t1) | nondigit_bits(r2, t2) | nondigit_bits(r3, t3);
                                    const uint32_t lo8 = __umul24(four_digits(t2), 10000u) + four_digits(t3);
                                    const uint32_t hi8 = __umul24(four_digits(t0), 10000u) + four_digits(t1);
                                    v_rec = v_range = (uint64_t)hi8 * 100000000ull + lo8;
                                } else {   // char(1): pass 1 adds byte 1 only after a backslash (:358-361), pass 2 always (:543-547)
                                    const uint32_t wrow = (fs >> 2) & (RW_RING_ROWS - 1u), sh = fs & 3u;
                                    const uint32_t *wq = reinterpret_cast<const uint32_t *>(zdw_lds + ring_l + wrow * 256u);
                                    const uint32_t x = __builtin_amdgcn_alignbyte(wq[64], wq[0], sh);
                                    const uint32_t b0 = x & 0xffu, b1 = len >= 2u ? ((x >> 8) & 0xffu) : 0u;
                                    const int64_t c0 = (int8_t)b0;
                                    const int64_t hi8 = (int64_t)((int)(int8_t)b1 * 256);
                                    v_range = (uint64_t)(b0 == '\\' ? c0 + hi8 : c0);
                                    v_rec = (uint64_t)(c0 + hi8);
                                }
                                if ((junk == 0u) & (v_range > 0)) range_update(a, t, col, (meta >> 2) & 0x3fffu, v_range);
                                n_rec = v_rec;
                            }
                            slowm |= BAL(junk != 0u);                       // signs, blanks, letters: the general form
                        }
                        if (!nb_n) off_n = recoff;
                        const uint32_t so = recoff - off_n;
                        const uint64_t rv = lane_in(slowm) ? 0ull : n_rec;
                        uint32_t *sp = reinterpret_cast<uint32_t *>(zdw_lds + stage_n + lane * RW_STAGE_N + so);
                        sp[0] = (uint32_t)rv; sp[1] = (uint32_t)(rv >> 32);
                        nb_n = so + 8u;
                    }
                    RW_TICK(4);          // 4: the field's own work
                    defer(slowm, gbase + fs, len, col);
                    // a lane whose cursor left its ring: the bytes it needs now (waiting for them)
                    {
                        const uint64_t rsm = actm & ~deadm & ~carefulm & ~okm;
                        if (rsm) resync(rsm);
                    }
                    RW_TICK(6);          // 6: queueing for the general form, rings refilled at once
                    // ---- the next refill: a lane that runs short starts a round, every lane with room joins ----
                    if (!pendm) {
                        const uint64_t fm = ~deadm & ~carefulm & okm & actm;
                        if (fm & BAL(avail < RW_ISSUE)) {
                            pendm = fm & BAL(avail <= RW_ROOM);
                            if (pendm & BAL(fill + 48u > last16)) {
                                // (the image ends inside the unit: its last pieces clamped, the ordinary way)
                                if (lane_in(pendm)) {
                                    rw_u32x4 v[4];
                                    load_unit(fill, v);
                                    store_unit(fill, v);
                                    fill += RW_UNIT; avail += (int32_t)RW_UNIT;
                                }
                                pendm = 0;
                            } else {
                                if (lane_in(pendm)) rw_unit_request(fill, imgb);
                                vm_seq += 4u;
                                pend_mark = vm_seq;
                            }
                        }
                    }
                    RW_TICK(7);          // 7: the next refill requested
                };
                RW_TICK(8);              // 8: staged records written out
                step(std::integral_constant<int, 0>{}, col0);
                if (col0 + 1 < ncols) {
                    step(std::integral_constant<int, 1>{}, col0 + 1);
                    if (col0 + 2 < ncols) {
                        step(std::integral_constant<int, 2>{}, col0 + 2);
                        if (col0 + 3 < ncols) step(std::integral_constant<int, 3>{}, col0 + 3);
                    }
                }
            }
            RW_TICK(9);
            // ---- the row's end ----
            drain();
            flush(stage_n, RW_STAGE_N, 0u, off_n, nb_n);
            flush_text(t_cnt);
            flush(stage_t, RW_STAGE_T, 16u * ((t_flushed >> 2) % 3u), t_base + 4u * t_flushed, 4u * (t_cnt - t_flushed));
            if (act) {
                P1Status *const stp = RW_COLD(st);
                if (row < nstore) {
                    // a row of ncols fields ends exactly at its terminator (ConvertToZDW.cpp:336-337): with fewer the walk ran past it,
                    // with more it stopped short
                    if (lane_in(deadm) || gbase + cursor - 1u != T) atomicMin(&stp->bad_row, (uint32_t)row);
                    if (row + 1 == a.row_limit) stp->split_pos = T + 1;             // the next block starts after this row
                }
                const uint64_t rl = T + 1 - S;                                       // rows that can grow the reference's line buffer (m_LongestLine)
                if (T >= S && T < RW_COLD(hi) && rl >= FIRST_ROW_CAP - 1) {
                    const uint32_t li = atomicAdd(&stp->long_count, 1u);
                    if (li < RW_COLD(long_cap)) { LongRow *lr = RW_COLD(long_rows); lr[li].start = S; lr[li].len = rl; }
                }
            }
        }
        if (pendm) { rw_wait_loads(0u); pendm = 0; }      // (a refill still in flight belongs to rows this wave is done with: let it land in its registers)
    }
    RW_TICK(10);                 // 10: rows' ends, groups' starts
#ifdef ZDW_EXPERIMENTS
    if (lane == 0) {
        unsigned long long *d = RW_COLD(st)->dbg;
        atomicAdd(d + 0, dbg_t0); atomicAdd(d + 1, dbg_t1); atomicAdd(d + 2, dbg_t2); atomicAdd(d + 3, dbg_t3); atomicAdd(d + 4, dbg_t4); atomicAdd(d + 5, dbg_t5);
        atomicAdd(d + 6, dbg_t6); atomicAdd(d + 7, dbg_t7); atomicAdd(d + 8, dbg_t8); atomicAdd(d + 9, dbg_t9); atomicAdd(d + 10, dbg_t10); atomicAdd(d + 11, dbg_t11);
    }
#endif
    retire_ids(ids, RW_COLD(uniq_len), RW_COLD(id_cap));
    {
        uint64_t *const dq = RW_COLD(dq);
        const uint32_t dq_cap = RW_COLD(dq_cap);
        for (uint32_t q = dq_cur + lane; q < dq_end; q += 64)
            if (q < dq_cap) dq[2ull * q + 1] = ~0ull;
    }
    __syncthreads();
    if (new_entries && lane == 0) {
        P1Status *const stp = RW_COLD(st);
        const uint32_t total = atomicAdd(&stp->entries, new_entries) + new_entries;
        if (total > RW_COLD(max_entries)) stp->overflow = 1;
    }
    // one flush of the column tables per workgroup
    {
        const uint8_t *ck = RW_COLD(col_kind);
        const int32_t *cn = RW_COLD(col_numidx);
        uint8_t *cs = RW_COLD(col_seen);
        uint64_t *cmn = RW_COLD(col_min), *cmx = RW_COLD(col_max);
        for (uint32_t c = threadIdx.x; c < ncols; c += blockDim.x) {
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

size_t pass1_rows_lds_bytes(uint32_t ncols, uint32_t nnum, uint32_t wpb) { return rw_layout(ncols, nnum, wpb).total; }

void launch_pass1_rows(const Pass1Args &a, const RowsArgs &ra, hipStream_t s)
{
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    typedef void (*Fn)(const Pass1Args, const RowsArgs);
#ifdef ZDW_EXPERIMENTS
    // (the look-up-only variant exists for one measurement of the profiling tools: DESIGN.md 4.1b)
    const Fn kfn = ra.find_only ? (a.track_first ? k_pass1_rows<true, true> : k_pass1_rows<false, true>)
                                : (a.track_first ? k_pass1_rows<true, false> : k_pass1_rows<false, false>);
#else
    const Fn kfn = a.track_first ? k_pass1_rows<true, false> : k_pass1_rows<false, false>;
#endif
    uint32_t wpb = RW_WPB_MAX;
    if (const char *e = test_hook("ZDW_P1_ROWS_WPB")) wpb = (uint32_t)atoi(e);
    if (wpb < 1) wpb = 1;
    if (wpb > RW_WPB_MAX) wpb = RW_WPB_MAX;
    while (wpb > 1 && rw_layout(a.ncols, a.nnum, wpb).total > 160u * 1024u) --wpb;
    const size_t lds = rw_layout(a.ncols, a.nnum, wpb).total;
    if (lds > 64u * 1024u) (void)hipFuncSetAttribute((const void *)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    uint32_t per_cu = (uint32_t)((160u * 1024u) / (((lds + 1279) / 1280) * 1280));
    if (per_cu < 1) per_cu = 1;
    hipLaunchKernelGGL(kfn, dim3((uint32_t)cus * per_cu), dim3(64 * wpb), lds, s, a, ra);
}

}  // namespace zdw
