// k_pass1.hip — pass 1 of the ZDW block encode on gfx950: row discovery, tokenising, string
// dedupe into the block dictionary table, integer parse, per-column ranges.
//
// Replaces the per-row loop of ConvertToZDW::parseInput (cplusplus/ConvertToZDW.cpp:329-414)
// together with GetNextRow (getnextrow.cpp:26-84), GetDataRow/get_next_column
// (ConvertToZDW.cpp:265-323,1048-1067) and Dictionary::insert (dictionary.cpp:31-51).
//
// Work decomposition (wave64, one wave per workgroup):
//   * the TSV image is cut into 16 KiB spans; a wave visit owns every row that STARTS in its span
//     and follows the last of them past the span end;
//   * the wave streams 256-byte rounds (one coalesced dword per lane); `v_cmp` ballots give
//     wave-wide bitmaps of tabs / newlines / backslashes in SGPRs, so row starts, field ordinals
//     and boundaries are mbcnt / scalar bit arithmetic, no per-byte loops.  Rounds that lie inside
//     the span and hold no row terminator (the common case for wide rows) take a short path;
//   * every field boundary becomes a 16-byte descriptor (end position, column, row) in a per-wave
//     LDS ring; a field starts where the previous descriptor ends.  Whenever 64 are pending the wave
//     processes them one per lane (hash + table insert, strtoull, char(1)), so field work runs at
//     full lane occupancy whatever the field length distribution is;
//   * pass 2 never re-reads the text: each field leaves a fixed-width record (u32 table slot id
//     or u64 parsed value) in rec[row][column].
#include "zdw_dev.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"

namespace zdw {

constexpr uint32_t QCAP = 384;   // ring entries per wave: <= 63 pending + 1 anchor + <= 256 new per round, + the predecessor of the head
constexpr uint32_t P1_MAX_WAVES = 256 * 16;

// ------------------------------------------------------------------------------------------
// byte classes of one round as wave-wide lane masks (bit l of mask[k] <-> byte k of lane l's dword)
// ------------------------------------------------------------------------------------------
struct RawMasks {
    uint64_t tab[4], nl[4], bs[4];
    uint64_t nul;
};

__device__ __forceinline__ void raw_classify(uint32_t w, RawMasks &r)
{
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = w & (0xffu << (8 * k));
        r.tab[k] = __ballot(b == ((uint32_t)'\t' << (8 * k)));
        r.nl[k] = __ballot(b == ((uint32_t)'\n' << (8 * k)));
        r.bs[k] = __ballot(b == ((uint32_t)'\\' << (8 * k)));
    }
    r.nul = __ballot(((w - 0x01010101u) & ~w & 0x80808080u) != 0u);   // some byte of the dword is zero
}

// Drops escaped tabs/newlines: special iff an even number of backslashes precedes.
// bs_carry: the byte before this round is a backslash (in/out).
__device__ __forceinline__ void resolve_escapes(const uint8_t *in, uint64_t lo, uint64_t pos, uint32_t lane, RawMasks &r, uint64_t &bs_carry)
{
    uint64_t prevbs[4];
    prevbs[0] = (r.bs[3] << 1) | bs_carry;
    prevbs[1] = r.bs[0]; prevbs[2] = r.bs[1]; prevbs[3] = r.bs[2];
    bs_carry = r.bs[3] >> 63;
    uint64_t cand = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) cand |= (r.tab[k] | r.nl[k]) & prevbs[k];
    if (cand) {   // rare: count the run exactly (bounded by the block start)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t ck = (r.tab[k] | r.nl[k]) & prevbs[k];
            bool esc = false;
            if (__builtin_amdgcn_inverse_ballot_w64(ck)) esc = backslashes_before(in, lo, pos + 4ull * lane + k) & 1u;
            const uint64_t e = __ballot(esc);
            r.tab[k] &= ~e;
            r.nl[k] &= ~e;
        }
    }
}

// lanes whose byte k lies at a round-relative position in [rel_lo, rel_hi)
__device__ __forceinline__ uint64_t lanes_in(uint32_t rel_lo, uint32_t rel_hi, int k)
{
    const int l_lo = rel_lo > (uint32_t)k ? (int)((rel_lo - k + 3) >> 2) : 0;
    const int l_hi = rel_hi > (uint32_t)k ? (int)((rel_hi - k + 3) >> 2) : 0;
    return mask_lt(l_hi) & ~mask_lt(l_lo);
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
        uint64_t pos = (n - 1) & ~(uint64_t)(ROUND_BYTES - 1);
        for (;;) {
            const uint32_t w = load_round_dword(in, pos, lane, n);
            uint64_t bs_carry = (pos > lo && in[pos - 1] == '\\') ? 1ull : 0ull;
            RawMasks r;
            raw_classify(w, r);
            resolve_escapes(in, lo, pos, lane, r, bs_carry);
            const uint32_t rel_lo = lo > pos ? (uint32_t)(lo - pos) : 0u;
            const uint32_t rel_hi = n - pos < ROUND_BYTES ? (uint32_t)(n - pos) : ROUND_BYTES;
            int best = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int hb = highest_bit(r.nl[k] & lanes_in(rel_lo, rel_hi, k));
                if (hb >= 0 && 4 * hb + k > best) best = 4 * hb + k;
            }
            if (best >= 0) { result = pos + (uint64_t)best + 1; break; }
            if (pos <= lo) break;          // pos is a multiple of 256, so pos > lo implies pos >= 256
            pos -= ROUND_BYTES;
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
// HBM-bound streaming kernel: 1 byte read per input byte, ~16 VALU + ~30 SALU per 256 bytes.
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
        uint32_t count = 0;
        uint64_t nul = 0;
        if (ss < se) {
            uint64_t prev_term = ((ss == lo) || is_row_terminator(in, lo, ss - 1)) ? 1ull : 0ull;
            const uint64_t pos0 = ss & ~(uint64_t)(ROUND_BYTES - 1);
            uint64_t bs_carry = (pos0 > lo && in[pos0 - 1] == '\\') ? 1ull : 0ull;
            uint32_t nq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) nq[q] = load_round_dword(in, pos0 + (uint64_t)q * ROUND_BYTES, lane, n);
            for (uint64_t pos = pos0; pos < se; pos += 4 * ROUND_BYTES) {
                // the next four rounds are requested before the current four are looked at
                uint32_t wq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { wq[q] = nq[q]; nq[q] = load_round_dword(in, pos + (uint64_t)(q + 4) * ROUND_BYTES, lane, n); }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint64_t rpos = pos + (uint64_t)q * ROUND_BYTES;
                    if (rpos >= se) break;
                    RawMasks r;
                    raw_classify(wq[q], r);
                    resolve_escapes(in, lo, rpos, lane, r, bs_carry);
                    if (rpos >= ss && rpos + ROUND_BYTES <= se) {
                        // interior round: every byte counts
                        const uint64_t pt0 = (r.nl[3] << 1) | prev_term;
                        count += popc64(pt0 & ~r.nl[0]) + popc64(r.nl[0] & ~r.nl[1]) + popc64(r.nl[1] & ~r.nl[2]) + popc64(r.nl[2] & ~r.nl[3]);
                        nul |= r.nul;
                        prev_term = r.nl[3] >> 63;
                    } else {
                        // first / last round of the span: only positions in [ss, se) exist
                        const uint32_t rel_lo = ss > rpos ? (uint32_t)(ss - rpos) : 0u;
                        const uint32_t rel_hi = se - rpos < ROUND_BYTES ? (uint32_t)(se - rpos) : ROUND_BYTES;
                        uint64_t nl[4], vm[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) { vm[k] = lanes_in(rel_lo, rel_hi, k); nl[k] = r.nl[k] & vm[k]; }
                        uint64_t pt[4];
                        pt[0] = nl[3] << 1; pt[1] = nl[0]; pt[2] = nl[1]; pt[3] = nl[2];
                        if (prev_term) pt[rel_lo & 3] |= 1ull << (rel_lo >> 2);   // the byte before position rel_lo terminated a row
#pragma unroll
                        for (int k = 0; k < 4; ++k) count += popc64(pt[k] & ~nl[k] & vm[k]);
                        // a NUL only matters inside the span
                        const uint32_t w = wq[q];
                        bool z = false;
#pragma unroll
                        for (int k = 0; k < 4; ++k) z |= __builtin_amdgcn_inverse_ballot_w64(vm[k]) && ((w >> (8 * k)) & 0xffu) == 0u;
                        nul |= __ballot(z);
                        prev_term = nl[3] >> 63;
                    }
                }
            }
        }
        if (lane == 0) {
            span_starts[si] = count;
            if (nul) st->nul_seen = 1;
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
// field-level work
// ------------------------------------------------------------------------------------------
// C-locale strtoull(s, NULL, 10) on a byte range (ConvertToZDW.cpp:385,564): the
// general form, one 8-byte window load per 8 characters.
__device__ inline uint64_t parse_u64_general(const uint8_t *p, uint32_t len, const uint8_t *end)
{
    uint64_t win = 0, v = 0;
    bool neg = false, any = false, ovf = false;
    uint32_t state = 0;     // 0 leading space, 1 digits
    for (uint32_t i = 0; i < len; ++i) {
        if ((i & 7u) == 0) win = load_upto8(p + i, len - i < 8 ? len - i : 8u, end);
        const uint32_t b = (uint32_t)(win >> (8 * (i & 7u))) & 0xffu;
        if (state == 0) {
            if (b == ' ' || (b >= 9 && b <= 13)) continue;
            state = 1;
            if (b == '+' || b == '-') { neg = b == '-'; continue; }
        }
        const uint32_t d = b - '0';
        if (d > 9) break;
        any = true;
        if (v > 0xFFFFFFFFFFFFFFFFull / 10 || (v == 0xFFFFFFFFFFFFFFFFull / 10 && d > 5)) ovf = true;
        else v = v * 10 + d;
    }
    if (!any) return 0;
    if (ovf) return 0xFFFFFFFFFFFFFFFFull;
    return neg ? (uint64_t)0 - v : v;
}

// eight ASCII digits (first character in the low byte) -> value, three multiplies
__device__ __forceinline__ uint32_t eight_digits(uint64_t v)
{
    v -= 0x3030303030303030ull;
    v = v * 10 + (v >> 8);
    v = (((v & 0x000000FF000000FFull) * 0x000F424000000064ull) + (((v >> 16) & 0x000000FF000000FFull) * 0x0000271000000001ull)) >> 32;
    return (uint32_t)v;
}
__device__ __forceinline__ bool all_digits(uint64_t w)
{
    return (((w + 0x4646464646464646ull) | (w - 0x3030303030303030ull)) & 0x8080808080808080ull) == 0ull;
}
// `len` (1..8) characters in the low bytes -> eight characters, '0'-padded in front
__device__ __forceinline__ uint64_t pad_digits(uint64_t w, uint32_t len)
{
    return len >= 8 ? w : ((w << (8 * (8 - len))) | (0x3030303030303030ull >> (8 * len)));
}

// 1..16 plain digits take the SWAR route (two window loads, no per-character loop); anything else
// (sign, blanks, junk, 17+ characters) goes through the general parser.  Same result either way.
__device__ inline uint64_t parse_u64(const uint8_t *p, uint32_t len, const uint8_t *end)
{
    if (len <= 8) {
        const uint64_t w = pad_digits(load_upto8(p, len, end), len);
        if (all_digits(w)) return eight_digits(w);
    } else if (len <= 16) {
        const uint64_t w1 = pad_digits(load_upto8(p, len - 8, end), len - 8);
        const uint64_t w2 = load_upto8(p + len - 8, 8, end);
        if (all_digits(w1) && all_digits(w2)) return (uint64_t)eight_digits(w1) * 100000000ull + eight_digits(w2);
    }
    return parse_u64_general(p, len, end);
}

struct RangeSink {           // where per-column ranges of this wave accumulate
    uint64_t *mn, *mx;       // indexed by numeric index (LDS) or column (global)
    uint8_t *seen;           // indexed by column
    bool lds;
};

__device__ __forceinline__ void range_update(const RangeSink &r, uint32_t idx, uint64_t v)
{
    // unsigned 64-bit compare like the reference (ConvertToZDW.cpp:362-373,386-397).  The plain reads
    // only filter: min only falls and max only rises, so a stale value can cost a needless atomic,
    // never lose an update.
    if (v < r.mn[idx]) atomicMin(reinterpret_cast<unsigned long long *>(&r.mn[idx]), (unsigned long long)v);
    if (v > r.mx[idx]) atomicMax(reinterpret_cast<unsigned long long *>(&r.mx[idx]), (unsigned long long)v);
}

__device__ inline void process_field(const Pass1Args &a, const RangeSink &rs, uint64_t fs, uint32_t len, uint32_t col, uint32_t row,
                                     bool &is_new)
{
    const bool in_block = row < a.row_limit;
    if (!in_block && !(row == a.row_limit && col < a.partial_fields)) return;
    const uint8_t *p = a.in + fs;
    if (a.trim) while (len > 0 && p[len - 1] == ' ') --len;          // ConvertToZDW.cpp:295-313
    const uint8_t kind = a.col_kind[col];
    uint8_t *recp = a.rec + (uint64_t)row * a.rec_stride + a.col_recoff[col];
    if (kind == K_TEXT) {
        uint32_t id = 0;
        if (len) {                                                    // empty fields are skipped (:340-341)
            if (!rs.seen[col]) rs.seen[col] = 1;
            if (!(a.debug_skip & 2u))
                id = dict_insert(a.in, a.n, a.extra, a.hi, a.slots, a.slot_len, a.slot_mask, a.st, fs, len, a.trim, is_new);
        }
        if (in_block && !(a.debug_skip & 8u)) *reinterpret_cast<uint32_t *>(recp) = id;
        return;
    }
    uint64_t v_range = 0, v_rec = 0;
    if (len) {
        if (kind == K_INT) {
            v_range = v_rec = (a.debug_skip & 4u) ? 1 : parse_u64(p, len, a.in + a.n);
        } else {   // char(1): pass 1 adds byte 1 only after a backslash (:358-361), pass 2 always (:543-547)
            const int64_t c0 = (int8_t)p[0];
            const int64_t hi8 = (int64_t)((int)(int8_t)(len >= 2 ? p[1] : 0) * 256);
            v_range = (uint64_t)(p[0] == '\\' ? c0 + hi8 : c0);
            v_rec = (uint64_t)(c0 + hi8);
        }
        if (v_range > 0) range_update(rs, rs.lds ? (uint32_t)a.col_numidx[col] : col, v_range);
    }
    if (in_block && !(a.debug_skip & 8u)) *reinterpret_cast<uint64_t *>(recp) = v_rec;
}

// ring entry: x = end[31:0], y = end[39:32] | col << 8, z = row.  The field it closes starts one
// byte after the end recorded by the entry before it (anchors and blank-line terminators are kept
// in the ring as COL_SKIP entries for exactly that purpose).
__device__ __forceinline__ uint4 make_entry(uint64_t end, uint32_t col, uint32_t row)
{
    uint4 e;
    e.x = (uint32_t)end;
    e.y = (uint32_t)((end >> 32) & 0xffu) | (col << 8);
    e.z = row;
    e.w = 0;
    return e;
}
// same, from the low word and bits 39:32 of the position (a round never straddles a 4 GiB line,
// so the low word of pos + 4*lane + k is a plain 32-bit add)
__device__ __forceinline__ uint4 make_entry32(uint32_t end_lo, uint32_t end_hi8, uint32_t col, uint32_t row)
{
    uint4 e;
    e.x = end_lo;
    e.y = end_hi8 | (col << 8);
    e.z = row;
    e.w = 0;
    return e;
}
__device__ __forceinline__ uint32_t ring_next(uint32_t i) { return i + 1 == QCAP ? 0u : i + 1; }
__device__ __forceinline__ uint64_t entry_end(const uint4 &e) { return ((uint64_t)(e.y & 0xffu) << 32) | e.x; }

// returns how many new dictionary entries the batch created
__device__ __forceinline__ uint32_t consume(const Pass1Args &a, const RangeSink &rs, const uint4 *ring, uint32_t head, uint32_t count,
                                            uint32_t lane)
{
    bool is_new = false;
    if (lane < count) {
        const uint4 e = ring[(head + lane) % QCAP];
        const uint32_t col = e.y >> 8;
        if (col != COL_SKIP && !(a.debug_skip & 1u)) {
            const uint4 pe = ring[(head + lane + QCAP - 1) % QCAP];
            // positions are 40-bit; the predecessor's end + 1 is where this field begins
            const uint64_t fs = (entry_end(pe) + 1) & POS_MASK;
            process_field(a, rs, fs, (uint32_t)(entry_end(e) - fs), col, e.z, is_new);
        }
    }
    return popc64(__ballot(is_new));
}

size_t pass1_lds_bytes(uint32_t ncols, uint32_t nnum, bool use_lds)
{
    size_t b = QCAP * sizeof(uint4);
    if (use_lds) b += (size_t)nnum * 16 + ((ncols + 15) & ~15u);
    return b;
}

__global__ __launch_bounds__(64) void k_pass1(const Pass1Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_raw[];
    uint4 *ring = reinterpret_cast<uint4 *>(lds_raw);
    const uint32_t lane = lane_id();
    RangeSink rs;
    rs.lds = a.use_lds_ranges != 0;
    if (rs.lds) {
        rs.mn = reinterpret_cast<uint64_t *>(lds_raw + QCAP * sizeof(uint4));
        rs.mx = rs.mn + a.nnum;
        rs.seen = reinterpret_cast<uint8_t *>(rs.mx + a.nnum);
        for (uint32_t i = lane; i < a.nnum; i += 64) { rs.mn[i] = ~0ull; rs.mx[i] = 0; }
        for (uint32_t i = lane; i < a.ncols; i += 64) rs.seen[i] = 0;
    } else {
        rs.mn = a.col_min; rs.mx = a.col_max; rs.seen = a.col_seen;
    }
    __syncthreads();

    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running (kept small enough for % QCAP to stay exact)
    uint32_t new_entries = 0;          // dictionary inserts of this wave not yet added to st->entries

    for (uint32_t si = blockIdx.x; si < a.nspans; si += gridDim.x) {
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

        // wave-uniform running state of the open row
        const bool at_row_start = (ss == a.lo) || is_row_terminator(a.in, a.lo, ss - 1);
        uint64_t st_row_start = ss;
        uint32_t st_tabs = 0, st_rows = 0;
        bool st_ignore = !at_row_start;      // the row open at ss began in an earlier span
        uint64_t prev_term = at_row_start ? 1ull : 0ull;
        uint64_t pos = ss & ~(uint64_t)(ROUND_BYTES - 1);
        uint64_t bs_carry = (pos > a.lo && a.in[pos - 1] == '\\') ? 1ull : 0ull;
        bool first = true, done = false;

        // anchor: the first field of this visit starts at ss
        if (lane == 0) ring[q_tail % QCAP] = make_entry((ss - 1) & POS_MASK, COL_SKIP, 0);
        q_tail += 1;

        // four rounds of input in flight: the dword of round r+4 is requested while round r is classified
        // Input arrives in groups of four rounds (1 KiB per wave): the next group is requested before the
        // current one is looked at, so its HBM latency hides behind four rounds of work.
        uint32_t n0 = load_round_dword(a.in, pos, lane, a.n);
        uint32_t n1 = load_round_dword(a.in, pos + 1 * ROUND_BYTES, lane, a.n);
        uint32_t n2 = load_round_dword(a.in, pos + 2 * ROUND_BYTES, lane, a.n);
        uint32_t n3 = load_round_dword(a.in, pos + 3 * ROUND_BYTES, lane, a.n);
        for (uint64_t gpos = pos; !done && gpos < a.hi; gpos += 4 * ROUND_BYTES) {
          const uint32_t c0 = n0, c1 = n1, c2 = n2, c3 = n3;
          n0 = load_round_dword(a.in, gpos + 4 * ROUND_BYTES, lane, a.n);
          n1 = load_round_dword(a.in, gpos + 5 * ROUND_BYTES, lane, a.n);
          n2 = load_round_dword(a.in, gpos + 6 * ROUND_BYTES, lane, a.n);
          n3 = load_round_dword(a.in, gpos + 7 * ROUND_BYTES, lane, a.n);
#pragma unroll 1
          for (uint32_t j = 0; j < 4 && !done; ++j) {
            pos = gpos + (uint64_t)j * ROUND_BYTES;
            if (pos >= a.hi) break;
            const uint32_t w = j == 0 ? c0 : (j == 1 ? c1 : (j == 2 ? c2 : c3));
            RawMasks r;
            raw_classify(w, r);
            resolve_escapes(a.in, a.lo, pos, lane, r, bs_carry);

            const bool interior = !first && pos + ROUND_BYTES <= se - 1;     // wholly inside the span, not its last byte
            if (interior && (r.nl[0] | r.nl[1] | r.nl[2] | r.nl[3]) == 0ull) {
                // ---------- short path: no row terminator in this round ----------
                if (prev_term) { st_rows += 1; st_row_start = pos; prev_term = 0; }
                const uint32_t before = mbcnt64(r.tab[0]) + mbcnt64(r.tab[1]) + mbcnt64(r.tab[2]) + mbcnt64(r.tab[3]);
                uint32_t col = st_ignore ? COL_SKIP : st_tabs + before;      // COL_SKIP + small stays >= ncols
                uint32_t qi = (q_tail + before) % QCAP;
                const uint32_t row = rowbase + st_rows - 1;
                const uint32_t lane_lo = (uint32_t)pos + 4u * lane, pos_hi8 = (uint32_t)(pos >> 32) & 0xffu;
                // this lane's tabs as a 4-bit set; lanes loop over their own (0.4 per lane on average)
                uint32_t mine = (__builtin_amdgcn_inverse_ballot_w64(r.tab[0]) ? 1u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(r.tab[1]) ? 2u : 0u) |
                                (__builtin_amdgcn_inverse_ballot_w64(r.tab[2]) ? 4u : 0u) | (__builtin_amdgcn_inverse_ballot_w64(r.tab[3]) ? 8u : 0u);
                while (mine) {
                    const uint32_t k = (uint32_t)__builtin_ctz(mine);
                    mine &= mine - 1;
                    ring[qi] = make_entry32(lane_lo + k, pos_hi8, col >= a.ncols ? COL_SKIP : col, row);
                    qi = ring_next(qi);
                    ++col;
                }
                const uint32_t total = popc64(r.tab[0]) + popc64(r.tab[1]) + popc64(r.tab[2]) + popc64(r.tab[3]);
                st_tabs += total;
                q_tail += total;
            } else {
                // ---------- general path: terminators, span edges ----------
                const uint32_t rel_lo = ss > pos ? (uint32_t)(ss - pos) : 0u;
                const uint32_t rel_hi = a.hi - pos < ROUND_BYTES ? (uint32_t)(a.hi - pos) : ROUND_BYTES;
                uint64_t term[4], tabm[4], keep[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    keep[k] = lanes_in(rel_lo, rel_hi, k);
                    term[k] = r.nl[k] & keep[k];
                    tabm[k] = r.tab[k] & keep[k];
                }
                // the visit ends with the first terminator at or after se - 1
                if (pos + ROUND_BYTES > se - 1) {
                    const uint32_t thr = se - 1 > pos ? (uint32_t)(se - 1 - pos) : 0u;
                    int cut = 1 << 30;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int lb = lowest_bit(term[k] & lanes_in(thr, ROUND_BYTES, k));
                        if (lb >= 0 && 4 * lb + k < cut) cut = 4 * lb + k;
                    }
                    if (cut < (1 << 30)) {
                        done = true;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            keep[k] &= lanes_in(0, (uint32_t)cut + 1, k);
                            term[k] &= keep[k];
                            tabm[k] &= keep[k];
                        }
                    }
                }

                uint64_t pt[4];
                pt[0] = term[3] << 1; pt[1] = term[0]; pt[2] = term[1]; pt[3] = term[2];
                if (first) {
                    if (prev_term) pt[rel_lo & 3] |= 1ull << (rel_lo >> 2);
                    first = false;
                } else pt[0] |= prev_term;

                uint64_t bnd[4], blank[4], start[4];
                uint32_t nbnd_before = 0, nstart_before = 0, ntab_before = 0, nbnd_total = 0;
                int tpos = -1;
                const uint64_t lt = mask_lt((int)lane);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    blank[k] = term[k] & pt[k];                // empty line: terminator right after a terminator
                    bnd[k] = tabm[k] | term[k];
                    start[k] = pt[k] & ~term[k] & keep[k];
                    nbnd_before += mbcnt64(bnd[k]);
                    nstart_before += mbcnt64(start[k]);
                    ntab_before += mbcnt64(tabm[k]);
                    nbnd_total += popc64(bnd[k]);
                    const int ht = highest_bit(term[k] & lt);
                    if (ht >= 0 && 4 * ht + k > tpos) tpos = 4 * ht + k;
                }

                // this lane's view of the open row at its first byte
                uint64_t my_row_start;
                uint32_t my_tabs;
                bool my_ignore;
                if (tpos >= 0) {
                    my_row_start = pos + (uint64_t)tpos + 1;
                    my_ignore = false;
                    const int tl = tpos >> 2, tk = tpos & 3;
                    uint32_t upto = 0;   // tabs at positions <= tpos
#pragma unroll
                    for (int k = 0; k < 4; ++k) upto += popc64(tabm[k] & mask_le(k <= tk ? tl : tl - 1));
                    my_tabs = ntab_before - upto;
                } else {
                    my_row_start = st_row_start;
                    my_ignore = st_ignore;
                    my_tabs = st_tabs + ntab_before;
                }
                uint32_t my_rows = st_rows + nstart_before;
                uint32_t qi = q_tail + nbnd_before;

#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint64_t p = pos + 4ull * lane + k;
                    if (__builtin_amdgcn_inverse_ballot_w64(start[k])) ++my_rows;
                    const bool is_term = __builtin_amdgcn_inverse_ballot_w64(term[k]);
                    const bool is_tab = __builtin_amdgcn_inverse_ballot_w64(tabm[k]);
                    const bool is_blank = __builtin_amdgcn_inverse_ballot_w64(blank[k]);
                    if (is_term || is_tab) {
                        const uint32_t row = rowbase + my_rows - 1;
                        uint32_t col = my_tabs;
                        if (is_blank || my_ignore || col >= a.ncols) col = COL_SKIP;
                        ring[qi % QCAP] = make_entry(p, col, row);
                        ++qi;
                        if (is_term && !is_blank && !my_ignore && row <= a.row_limit) {
                            if (my_tabs + 1 != a.ncols) atomicMin(&a.st->bad_row, row);       // ConvertToZDW.cpp:336-337
                            const uint64_t rl = p - my_row_start + 1;
                            if (rl >= FIRST_ROW_CAP - 1) {                                     // may grow m_LongestLine
                                const uint32_t li = atomicAdd(&a.st->long_count, 1u);
                                if (li < a.long_cap) { a.long_rows[li].start = my_row_start; a.long_rows[li].len = rl; }
                            }
                            if (row == a.row_limit) a.st->split_pos = my_row_start;
                        }
                    }
                    if (is_term) { my_tabs = 0; my_row_start = p + 1; my_ignore = false; }
                    else if (is_tab) ++my_tabs;
                }

                // lane 63 now holds the state at the end of the round
                st_row_start = bcast64(my_row_start, 63);
                st_tabs = bcast32(my_tabs, 63);
                st_rows = bcast32(my_rows, 63);
                st_ignore = bcast32(my_ignore ? 1u : 0u, 63) != 0;
                prev_term = term[3] >> 63;
                q_tail += nbnd_total;
            }

            __builtin_amdgcn_wave_barrier();      // one wave per workgroup: LDS operations of a wave execute in order
            while (q_tail - q_head >= 64) {
                new_entries += consume(a, rs, ring, q_head, 64, lane);
                q_head += 64;
            }
            if (q_head >= QCAP * 4096u) { q_head -= QCAP * 4096u; q_tail -= QCAP * 4096u; }   // keep (index % QCAP) exact
            __builtin_amdgcn_wave_barrier();
          }
        }
    }
    __syncthreads();
    if (q_tail != q_head) new_entries += consume(a, rs, ring, q_head, q_tail - q_head, lane);
    __syncthreads();
    if (new_entries && lane == 0) {
        const uint32_t total = atomicAdd(&a.st->entries, new_entries) + new_entries;
        if (total > a.max_entries) a.st->overflow = 1;
    }

    if (rs.lds) {   // one flush per wave
        for (uint32_t c = lane; c < a.ncols; c += 64) {
            if (a.col_kind[c] == K_TEXT) {
                if (rs.seen[c]) a.col_seen[c] = 1;
            } else {
                const uint32_t i = (uint32_t)a.col_numidx[c];
                if (rs.mx[i] != 0) {
                    atomicMin(reinterpret_cast<unsigned long long *>(&a.col_min[c]), (unsigned long long)rs.mn[i]);
                    atomicMax(reinterpret_cast<unsigned long long *>(&a.col_max[c]), (unsigned long long)rs.mx[i]);
                }
            }
        }
    }
}

void launch_pass1(const Pass1Args &a, hipStream_t s)
{
    if (!a.nspans) return;
    const size_t lds = pass1_lds_bytes(a.ncols, a.nnum, a.use_lds_ranges != 0);
    const uint32_t grid = a.nspans < P1_MAX_WAVES ? a.nspans : P1_MAX_WAVES;
    hipLaunchKernelGGL(k_pass1, dim3(grid), dim3(64), lds, s, a);
}

}  // namespace zdw
