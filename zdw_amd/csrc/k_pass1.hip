// k_pass1.hip — row scan and pass 1 of the ZDW block encode on gfx950: row discovery, tokenising,
// string dedupe into the block dictionary table, integer parse, per-column ranges.
//
// Replaces the per-row loop of ConvertToZDW::parseInput (cplusplus/ConvertToZDW.cpp:329-414)
// together with GetNextRow (getnextrow.cpp:26-84), GetDataRow/get_next_column
// (ConvertToZDW.cpp:265-323,1048-1067) and Dictionary::insert (dictionary.cpp:31-51).
//
// Work decomposition (wave64, one wave per workgroup):
//   * the TSV image is cut into 16 KiB spans; a wave visit owns every row that STARTS in its span
//     and follows the last of them past the span end (no row-length limit);
//   * the wave streams 1 KiB rounds: one coalesced 16-byte load per lane, the next round requested
//     before the current one is looked at.  Each lane turns its 16 bytes into 16-bit tab / newline /
//     backslash sets with SWAR compares (vector ALU only: the scalar unit is one per CU and was the
//     bottleneck of an earlier ballot-mask formulation);
//   * two wave prefix scans per round place every boundary: boundaries before the lane (ring slot)
//     and tabs since the last row terminator (column);
//   * every boundary becomes an 8-byte descriptor {end position, column, ends-a-row} in a per-wave
//     LDS ring; a field begins one byte after the previous descriptor's end and its row is the number
//     of row-ending descriptors before it.  Whenever 64 descriptors are pending the wave processes
//     them one per lane (zdw_field_dev.h), so field work runs at full lane occupancy whatever the
//     field-length distribution is;
//   * pass 2 never re-reads the text: each field leaves a fixed-width record (u32 table slot id
//     or u64 parsed value) in rec[row][column].
#include "zdw_dev.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"
#include "zdw_field_dev.h"

#include <cstdlib>

namespace zdw {

constexpr uint32_t WROUND = 1024;                 // bytes per wave round (16 per lane)
constexpr uint32_t QCAP = 384;                    // ring entries per wave; a round that does not fit is emitted in four 16-lane pieces (<= 256 each)
constexpr uint32_t P1_WPB = 2;                    // independent waves per workgroup (the CU takes 16 workgroups; 32 waves need 2 each)
constexpr uint32_t COL_FIELD_MASK = 0x7FFFFFu;    // descriptor: y = pos[39:32] | col << 8 | ends_row << 31
constexpr uint32_t COL_NONE = 0x7FFFFFu;          // boundary only (blank line, foreign row, anchor position)
constexpr uint32_t COL_ANCHOR = 0x7FFFFEu;        // x = row index of the rows that follow

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
    return *reinterpret_cast<const uint4 *>(in + base + off);
}

// bits i of a lane whose absolute position lp + i lies in [vlo, vhi)
__device__ __forceinline__ uint32_t valid16(uint64_t lp, uint64_t vlo, uint64_t vhi)
{
    const uint32_t lo_i = vlo > lp ? (vlo - lp < 16 ? (uint32_t)(vlo - lp) : 16u) : 0u;
    const uint32_t hi_i = vhi > lp ? (vhi - lp < 16 ? (uint32_t)(vhi - lp) : 16u) : 0u;
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

__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t lane, uint32_t &total)
{
    (void)lane;
    const uint32_t inc = wave_incl_sum(v);
    total = lane_value(inc, 63);
    return inc - v;
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
__global__ __launch_bounds__(64) void k_count_row_starts(const uint8_t *__restrict__ in, uint64_t n, uint64_t lo, uint64_t hi,
                                                          uint64_t span0, uint32_t nspans, uint32_t *__restrict__ span_starts,
                                                          P1Status *st)
{
    const uint32_t lane = lane_id();
    for (uint32_t si = blockIdx.x; si < nspans; si += gridDim.x) {
        const uint64_t s0 = (span0 + si) * (uint64_t)SPAN_BYTES;
        const uint64_t ss = s0 > lo ? s0 : lo;
        const uint64_t se = s0 + SPAN_BYTES < hi ? s0 + SPAN_BYTES : hi;
        uint32_t mine = 0, nul = 0;
        if (ss < se) {
            uint32_t prev_term = ((ss == lo) || is_row_terminator(in, lo, ss - 1)) ? 1u : 0u;
            const uint64_t pos0 = ss & ~(uint64_t)(WROUND - 1);
            uint32_t bs_carry = (pos0 > lo && in[pos0 - 1] == '\\') ? 1u : 0u;
            uint4 nxt = load_round16(in, pos0, lane, n);
            for (uint64_t pos = pos0; pos < se; pos += WROUND) {
                const uint4 v = nxt;
                nxt = load_round16(in, pos + WROUND, lane, n);            // next round in flight while this one is counted
                const uint64_t lp = pos + 16ull * lane;
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
            }
        }
        uint32_t total;
        wave_excl_sum(mine, lane, total);
        const uint64_t anynul = __ballot(nul != 0);
        if (lane == 0) {
            span_starts[si] = total;
            if (anynul) st->nul_seen = 1;
        }
    }
}

void launch_count_row_starts(const uint8_t *in, uint64_t n, uint64_t lo, uint64_t hi, uint64_t span0, uint32_t nspans,
                             uint32_t *span_starts, P1Status *st, hipStream_t s)
{
    if (!nspans) return;
    const uint32_t grid = nspans < 256u * 32u ? nspans : 256u * 32u;
    hipLaunchKernelGGL(k_count_row_starts, dim3(grid), dim3(64), 0, s, in, n, lo, hi, span0, nspans, span_starts, st);
}

// ------------------------------------------------------------------------------------------
// descriptor ring
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t ring_next(uint32_t i) { return i + 1 == QCAP ? 0u : i + 1; }
__device__ __forceinline__ uint64_t entry_pos(const uint2 &e) { return ((uint64_t)(e.y & 0xffu) << 32) | e.x; }

// Processes `count` (<= 64) descriptors starting at ring index `head`, one per lane.  row_cur = row
// index that applies at `head` (updated).  Returns how many new dictionary entries were created.
__device__ __forceinline__ uint32_t consume(const Pass1Args &a, const WaveCtx &w, uint32_t head, uint32_t count, uint32_t lane,
                                            uint32_t &row_cur)
{
    const uint2 *ring = reinterpret_cast<const uint2 *>(zdw_lds + w.ring);
    const bool live = lane < count;
    uint2 e = make_uint2(0u, COL_NONE << 8);
    uint32_t idx = head + lane;
    if (idx >= QCAP) idx -= QCAP;
    if (live) e = ring[idx];
    const uint32_t col = (e.y >> 8) & COL_FIELD_MASK;
    const bool ends_row = (e.y >> 31) != 0;
    // row index: the last anchor at or before this lane sets the base, row-ending descriptors after it count up
    const uint64_t am = __ballot(live && col == COL_ANCHOR);
    const uint64_t rm = __ballot(live && ends_row);
    const int an = highest_bit(am & mask_le((int)lane));
    const uint32_t anchored = (uint32_t)__shfl((int)e.x, an < 0 ? 0 : an, 64);
    const uint32_t row = (an >= 0 ? anchored : row_cur) + popc64(rm & mask_lt((int)lane) & ~mask_le(an));
    bool is_new = false;
    if (live && col < COL_ANCHOR) {
        if (ends_row && row <= a.row_limit) {
            if (col + 1 != a.ncols) atomicMin(&a.st->bad_row, row);              // ConvertToZDW.cpp:336-337
            if (row + 1 == a.row_limit) a.st->split_pos = entry_pos(e) + 1;      // the next block starts after this row
        }
        if (col < a.ncols && !(a.debug_skip & 1u)) {
            const uint2 pe = ring[idx ? idx - 1 : QCAP - 1];
            const uint64_t fs = (entry_pos(pe) + 1) & POS_MASK;                  // a field begins after its predecessor's end
            process_field(a, w, fs, (uint32_t)(entry_pos(e) - fs), col, row, is_new);
        }
    }
    // lanes past `count` hold no descriptor, so lane 63 sees the state after the last live one
    row_cur = lane_value(row + (ends_row ? 1u : 0u), 63);
    return popc64(__ballot(is_new));
}

// per-wave LDS: descriptor ring | byte ring + mirror | [column meta | min | max | seen]
size_t pass1_lds_bytes(uint32_t ncols, uint32_t nnum, bool use_lds)
{
    size_t b = QCAP * sizeof(uint2) + BYTE_RING + 16;
    if (use_lds) b += (((size_t)ncols * 4 + 7) & ~(size_t)7) + (size_t)nnum * 16 + ((ncols + 15) & ~15u);
    return (b + 15) & ~(size_t)15;
}

__global__ __launch_bounds__(64 * P1_WPB) void k_pass1(const Pass1Args a, uint32_t lds_per_wave)
{
    // every wave of the workgroup is an independent worker with its own slice of the LDS
    const uint32_t wb = (threadIdx.x >> 6) * lds_per_wave;
    const uint32_t wave_id = blockIdx.x * P1_WPB + (threadIdx.x >> 6), wave_count = gridDim.x * P1_WPB;
    const uint32_t lane = lane_id();
    WaveCtx w;
    w.ring = wb;
    w.bytes = wb + QCAP * sizeof(uint2);
    w.meta = w.bytes + BYTE_RING + 16;
    w.mn = w.meta + ((4 * a.ncols + 7u) & ~7u);           // 64-bit LDS atomics need 8-byte alignment
    w.mx = w.mn + 8 * a.nnum;
    w.seen = w.mx + 8 * a.nnum;
    w.cols_in_lds = a.use_lds_ranges != 0;
    w.lds_lo = w.lds_hi = 0;
    uint2 *ring = reinterpret_cast<uint2 *>(zdw_lds + w.ring);
    if (w.cols_in_lds) {
        for (uint32_t c = lane; c < a.ncols; c += 64) {
            *reinterpret_cast<uint32_t *>(zdw_lds + w.meta + 4 * c) = pack_col_meta(a.col_kind[c], (uint32_t)a.col_numidx[c] & 0x3fffu, a.col_recoff[c]);
            zdw_lds[w.seen + c] = 0;
        }
        for (uint32_t i = lane; i < a.nnum; i += 64) {
            reinterpret_cast<uint64_t *>(zdw_lds + w.mn)[i] = ~0ull;
            reinterpret_cast<uint64_t *>(zdw_lds + w.mx)[i] = 0;
        }
    }
    __syncthreads();

    uint32_t q_head = 0, q_tail = 0;   // ring indices in [0, QCAP)
    uint32_t q_count = 0;              // pending descriptors
    uint32_t row_cur = 0;              // row index at q_head
    uint32_t new_entries = 0;          // dictionary inserts of this wave not yet added to st->entries

    for (uint32_t si = wave_id; si < a.nspans; si += wave_count) {
        const uint32_t rowbase = a.span_rowbase[si];
        if (a.span_rowbase[si + 1] == rowbase) continue;          // no row starts in this span
        if (rowbase > a.row_limit) break;                         // later spans hold rows of later blocks only
        if (__hip_atomic_load(&a.st->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        if (new_entries >= 128) {        // publish this wave's inserts in bulk; stop everybody once the table is half full
            uint32_t seen_total = 0;
            if (lane == 0) seen_total = atomicAdd(&a.st->entries, new_entries) + new_entries;
            seen_total = bcast32(seen_total, 0);
            new_entries = 0;
            if (seen_total >= a.max_entries) { if (lane == 0) a.st->overflow = 1; break; }
        }
        const uint64_t s0 = (a.span0 + si) * (uint64_t)SPAN_BYTES;
        const uint64_t ss = s0 > a.lo ? s0 : a.lo;
        const uint64_t se = s0 + SPAN_BYTES < a.hi ? s0 + SPAN_BYTES : a.hi;

        // wave-uniform state of the open row
        const bool at_row_start = (ss == a.lo) || is_row_terminator(a.in, a.lo, ss - 1);
        uint64_t st_row_start = ss;
        uint32_t st_tabs = 0;
        bool st_ignore = !at_row_start;      // the row open at ss began in an earlier span and belongs to it
        uint32_t prev_term = at_row_start ? 1u : 0u;
        const uint64_t pos0 = ss & ~(uint64_t)(WROUND - 1);
        uint32_t bs_carry = (pos0 > a.lo && a.in[pos0 - 1] == '\\') ? 1u : 0u;
        bool done = false;

        // anchor pair: the row index that applies from here on, then the position the first field starts after
        if (lane == 0) {
            ring[q_tail] = make_uint2(rowbase - 1u + (at_row_start ? 1u : 0u), COL_ANCHOR << 8);
            ring[ring_next(q_tail)] = make_uint2((uint32_t)(ss - 1), ((uint32_t)((ss - 1) >> 32) & 0xffu) | (COL_NONE << 8));
        }
        q_tail = ring_next(ring_next(q_tail));
        q_count += 2;

        uint4 nxt = load_round16(a.in, pos0, lane, a.n);
        for (uint64_t pos = pos0; !done && pos < a.hi; pos += WROUND) {
            const uint4 v = nxt;
            nxt = load_round16(a.in, pos + WROUND, lane, a.n);            // next round in flight
            const uint64_t lp = pos + 16ull * lane;
            // keep the round in the LDS byte ring (two rounds deep): field work reads its bytes from there
            {
                const uint32_t slot = (uint32_t)pos & (BYTE_RING - 1);            // 0 or 1024
                *reinterpret_cast<uint4 *>(zdw_lds + w.bytes + slot + 16u * lane) = v;
                if (slot == 0 && lane == 0) *reinterpret_cast<uint2 *>(zdw_lds + w.bytes + BYTE_RING) = make_uint2(v.x, v.y);   // mirror for reads that straddle the wrap
                if (pos == w.lds_hi) { w.lds_hi = pos + WROUND; if (w.lds_hi - w.lds_lo > BYTE_RING) w.lds_lo = w.lds_hi - BYTE_RING; }
                else { w.lds_lo = pos; w.lds_hi = pos + WROUND; }
                // bytes past the end of the image are re-read filler: never trust them
                if (w.lds_hi > a.n) w.lds_hi = a.n > w.lds_lo ? a.n : w.lds_lo;
            }
            Lane16 m;
            classify16(a.in, a.lo, a.n, lp, v, lane, bs_carry, m, true);

            // positions that exist for this visit: [ss, hi), cut after the first terminator at or past se - 1
            uint32_t vm = 0xffffu;
            const bool edge = pos < ss || pos + WROUND > se - 1;
            if (edge) {
                vm = valid16(lp, ss, a.hi);
                const uint32_t late = m.nl & vm & valid16(lp, se - 1, ~0ull);
                const uint64_t lb = __ballot(late != 0);
                if (lb) {
                    const int fl = lowest_bit(lb);
                    const uint32_t fb = lane_value((uint32_t)__builtin_ctz(late | 0x10000u), fl);
                    done = true;
                    if (lane > (uint32_t)fl) vm = 0;
                    else if (lane == (uint32_t)fl) vm &= (2u << fb) - 1u;
                }
            }
            const uint32_t nl = m.nl & vm, tab = m.tab & vm;

            // "previous byte terminated a row" for each of my bytes
            uint32_t pt = wave_shr1(nl >> 15) | ((nl << 1) & 0xffffu);
            if (prev_term) {
                const uint64_t first = pos > ss ? pos : ss;
                if (first >= lp && first < lp + 16) pt |= 1u << (uint32_t)(first - lp);
                if (pos >= ss) st_row_start = pos;      // the open row begins with this round
            }
            const uint32_t blank = nl & pt;              // empty line: terminator right after a terminator
            const uint32_t bnd = tab | nl;

            // wave scans: boundaries before this lane; tabs since the last terminator before this lane
            uint32_t nb_total;
            const uint32_t nb_before = wave_excl_sum((uint32_t)__builtin_popcount(bnd), lane, nb_total);
            const bool has_nl = nl != 0;
            const uint32_t tail = has_nl ? (uint32_t)__builtin_popcount(tab & ~((2u << (31u - (uint32_t)__builtin_clz(nl | 1u))) - 1u))
                                         : (uint32_t)__builtin_popcount(tab);
            uint32_t seg = (has_nl ? 0x80000000u : 0u) | tail;          // inclusive segmented sum, reset flag in bit 31
            {
                // same step structure as wave_incl_sum; (flag, count) pairs combine as: a set flag stops the carry
                auto fold = [&](uint32_t o) { if (!(seg & 0x80000000u)) seg = (o & 0x80000000u) | ((o & 0x7fffffffu) + seg); };
                fold(dpp_or_zero<DPP_ROW_SHR1, 0xf>(seg));
                fold(dpp_or_zero<DPP_ROW_SHR2, 0xf>(seg));
                fold(dpp_or_zero<DPP_ROW_SHR4, 0xf>(seg));
                fold(dpp_or_zero<DPP_ROW_SHR8, 0xf>(seg));
                fold(dpp_or_zero<DPP_ROW_BCAST15, 0xa>(seg));
                fold(dpp_or_zero<DPP_ROW_BCAST31, 0xc>(seg));
            }
            const uint32_t segx = wave_shr1(seg);                        // exclusive
            uint32_t col = (segx & 0x80000000u) ? (segx & 0x7fffffffu) : st_tabs + segx;
            const uint32_t end_tabs = lane_value((seg & 0x80000000u) ? (seg & 0x7fffffffu) : st_tabs + seg, 63);

            // first terminator of the round (wave-uniform): only the row that was open when the round
            // began can be long enough to grow m_LongestLine
            const uint64_t tb = __ballot(has_nl);
            int ft_lane = -1;
            uint32_t ft_bit = 0;
            if (tb) {
                ft_lane = lowest_bit(tb);
                ft_bit = lane_value((uint32_t)__builtin_ctz(nl | 0x10000u), ft_lane);
                const uint32_t ft_blank = lane_value((blank >> ft_bit) & 1u, ft_lane);
                const uint64_t ft_pos = pos + 16ull * (uint32_t)ft_lane + ft_bit;
                const uint64_t rl = ft_pos - st_row_start + 1;
                if (!ft_blank && !st_ignore && rl >= FIRST_ROW_CAP - 1 && lane == 0) {
                    const uint32_t li = atomicAdd(&a.st->long_count, 1u);
                    if (li < a.long_cap) { a.long_rows[li].start = st_row_start; a.long_rows[li].len = rl; }
                }
            }

            // emit this lane's boundaries.  The ring always keeps the head's predecessor, so it can take
            // QCAP - 1 - pending new descriptors; a denser round is emitted as four 16-lane pieces (<= 256
            // boundaries each, which always fit an empty ring) with the ring drained in between.
            static_assert(QCAP >= 256 + 2, "a drained ring must take one 16-lane piece");
            const uint32_t lp_lo = (uint32_t)pos + 16u * lane, pos_hi8 = (uint32_t)(pos >> 32) & 0xffu;
            const bool lane_ign = st_ignore && (ft_lane < 0 || lane <= (uint32_t)ft_lane);
            const bool whole = q_count + nb_total <= QCAP - 1;
            const uint32_t b16 = lane_value(nb_before, 16), b32 = lane_value(nb_before, 32), b48 = lane_value(nb_before, 48);
            const uint32_t npieces = whole ? 1u : 4u;
            // One loop does both the emission (piece by piece) and the field work, so that the (large) field
            // code exists once: before a piece of a split round the ring is drained completely, otherwise
            // descriptors are taken 64 at a time.
            for (uint32_t piece = 0;; ++piece) {
                const bool drain = !whole && piece < npieces;
                __builtin_amdgcn_wave_barrier();      // one wave per worker: LDS operations of a wave execute in order
                while (drain ? q_count > 0 : q_count >= 64) {
                    const uint32_t cnt = q_count < 64 ? q_count : 64u;
                    new_entries += consume(a, w, q_head, cnt, lane, row_cur);
                    q_head += cnt;
                    if (q_head >= QCAP) q_head -= QCAP;
                    q_count -= cnt;
                }
                __builtin_amdgcn_wave_barrier();
                if (piece >= npieces) break;
                const bool active = whole || (lane >> 4) == piece;
                const uint32_t piece_base = whole ? 0u : (piece == 0 ? 0u : (piece == 1 ? b16 : (piece == 2 ? b32 : b48)));
                const uint32_t piece_end = whole ? nb_total : (piece == 0 ? b16 : (piece == 1 ? b32 : (piece == 2 ? b48 : nb_total)));
                const uint32_t piece_total = piece_end - piece_base;
                uint32_t qi = q_tail + (nb_before - piece_base);
                if (qi >= QCAP) qi -= QCAP;
                uint32_t bits = active ? bnd : 0u;
                while (bits) {
                    const uint32_t i = (uint32_t)__builtin_ctz(bits);
                    const uint32_t bit = bits & (0u - bits);
                    bits &= bits - 1;
                    const bool is_term = (nl & bit) != 0;
                    const bool ign = lane_ign && !(lane == (uint32_t)ft_lane && i > ft_bit);
                    uint32_t c = col < COL_ANCHOR ? col : COL_ANCHOR - 1;       // absurdly wide rows still count as "wrong"
                    uint32_t ends = is_term ? 0x80000000u : 0u;
                    if ((blank & bit) || ign) { c = COL_NONE; if (blank & bit) ends = 0; }
                    ring[qi] = make_uint2(lp_lo + i, pos_hi8 | (c << 8) | ends);
                    qi = ring_next(qi);
                    col = is_term ? 0u : col + 1;
                }
                q_tail += piece_total;
                if (q_tail >= QCAP) q_tail -= QCAP;
                q_count += piece_total;
            }

            // state at the end of the round
            st_tabs = end_tabs;
            if (tb) {
                st_ignore = false;
                const int lt_lane = highest_bit(tb);
                const uint32_t lt_bit = lane_value(31u - (uint32_t)__builtin_clz(nl | 1u), lt_lane);
                st_row_start = pos + 16ull * (uint32_t)lt_lane + lt_bit + 1;
            }
            prev_term = lane_value(nl >> 15, 63);
        }
    }
    __builtin_amdgcn_wave_barrier();
    if (q_count) new_entries += consume(a, w, q_head, q_count, lane, row_cur);
    __syncthreads();
    if (new_entries && lane == 0) {
        const uint32_t total = atomicAdd(&a.st->entries, new_entries) + new_entries;
        if (total > a.max_entries) a.st->overflow = 1;
    }

    if (w.cols_in_lds) {   // one flush per wave
        for (uint32_t c = lane; c < a.ncols; c += 64) {
            if (a.col_kind[c] == K_TEXT) {
                if (zdw_lds[w.seen + c]) a.col_seen[c] = 1;
            } else {
                const uint32_t i = (uint32_t)a.col_numidx[c];
                const uint64_t mn = reinterpret_cast<const uint64_t *>(zdw_lds + w.mn)[i], mx = reinterpret_cast<const uint64_t *>(zdw_lds + w.mx)[i];
                if (mx != 0) {
                    atomicMin(reinterpret_cast<unsigned long long *>(&a.col_min[c]), (unsigned long long)mn);
                    atomicMax(reinterpret_cast<unsigned long long *>(&a.col_max[c]), (unsigned long long)mx);
                }
            }
        }
    }
}

void launch_pass1(const Pass1Args &a, hipStream_t s)
{
    if (!a.nspans) return;
    const size_t lds_wave = (pass1_lds_bytes(a.ncols, a.nnum, a.use_lds_ranges != 0) + 15) & ~(size_t)15;
    const size_t lds = lds_wave * P1_WPB;
    // Waves walk the spans with a grid stride, so every wave carries the same share of the block: the grid
    // must not exceed what is resident at once, or the surplus waves run alone after the others finish.
    int per_cu = 0, dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_pass1, 64 * P1_WPB, lds) != hipSuccess || per_cu < 1) per_cu = 4;
    const size_t by_lds = (160u * 1024u) / (((lds + 1279) / 1280) * 1280);     // LDS is granted in 1280-byte granules
    if ((size_t)per_cu > by_lds && by_lds >= 1) per_cu = (int)by_lds;
    if (per_cu > 16) per_cu = 16;                                               // workgroup slots per CU
    uint32_t max_blocks = (uint32_t)cus * (uint32_t)per_cu;
    if (const char *e = getenv("ZDW_P1_WAVES_PER_CU")) max_blocks = (uint32_t)cus * (uint32_t)atoi(e) / P1_WPB;   // tuning experiments
    const uint32_t want_blocks = (a.nspans + P1_WPB - 1) / P1_WPB;
    const uint32_t grid = want_blocks < max_blocks ? want_blocks : (max_blocks ? max_blocks : 1u);
    hipLaunchKernelGGL(k_pass1, dim3(grid), dim3(64 * P1_WPB), lds, s, a, (uint32_t)lds_wave);
}

}  // namespace zdw
