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
//     wave-wide bitmaps of tabs / newlines / backslashes, so row starts, field ordinals and field
//     starts are mbcnt / find-highest-bit arithmetic on SGPR masks, no per-byte loops;
//   * every field boundary becomes a 16-byte descriptor in a per-wave LDS ring; whenever 64 are
//     pending the wave processes them one per lane (hash + table insert, strtoull, char(1)),
//     so field work runs at full lane occupancy whatever the field length distribution is;
//   * pass 2 never re-reads the text: each field leaves a fixed-width record (u32 table slot id
//     or u64 parsed value) in rec[row][column].
#include "zdw_dev.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"

namespace zdw {

constexpr uint32_t QCAP = 512;            // ring entries per wave (<= 63 pending + <= 256 new per round)
constexpr uint32_t P1_MAX_WAVES = 256 * 12;

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
            const uint32_t w = load_dword_guarded(in, pos + 4ull * lane, n);
            bool bs_carry = pos > lo && in[pos - 1] == '\\';
            RoundMasks m;
            classify_round(in, lo, lo, n, pos, w, lane, bs_carry, m);
            int best = -1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int hb = highest_bit(m.nl[k]);
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
            bool prev_term = (ss == lo) || is_row_terminator(in, lo, ss - 1);
            uint64_t pos = ss & ~(uint64_t)(ROUND_BYTES - 1);
            bool bs_carry = pos > lo && in[pos - 1] == '\\';
            bool first = true;
            for (; pos < se; pos += ROUND_BYTES) {
                const uint32_t w = load_dword_guarded(in, pos + 4ull * lane, n);
                RoundMasks m;
                classify_round(in, lo, ss, se, pos, w, lane, bs_carry, m);
                nul |= m.nul;
                uint64_t pt[4];
                pt[0] = m.nl[3] << 1; pt[1] = m.nl[0]; pt[2] = m.nl[1]; pt[3] = m.nl[2];
                if (first) {
                    const uint32_t fv = (uint32_t)(ss - pos);
                    if (prev_term) pt[fv & 3] |= 1ull << (fv >> 2);
                    first = false;
                } else if (prev_term) pt[0] |= 1ull;
                // positions in [ss, se) that hold a real byte
                const uint64_t rel_lo = ss > pos ? ss - pos : 0, rel_hi = se - pos < ROUND_BYTES ? se - pos : ROUND_BYTES;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // lanes l with rel_lo <= 4l+k < rel_hi
                    const int l_lo = rel_lo > (uint64_t)k ? (int)((rel_lo - k + 3) >> 2) : 0;
                    const int l_hi = rel_hi > (uint64_t)k ? (int)((rel_hi - k + 3) >> 2) : 0;   // exclusive
                    const uint64_t valid = mask_lt(l_hi) & ~mask_lt(l_lo);
                    count += popc64(pt[k] & ~m.nl[k] & valid);
                }
                prev_term = (m.nl[3] >> 63) & 1ull;
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
// C-locale strtoull(s, NULL, 10) on a byte range (ConvertToZDW.cpp:385,564)
__device__ inline uint64_t parse_u64(const uint8_t *p, uint32_t len)
{
    uint32_t i = 0;
    while (i < len && (p[i] == ' ' || (p[i] >= 9 && p[i] <= 13))) ++i;
    bool neg = false;
    if (i < len && (p[i] == '+' || p[i] == '-')) { neg = p[i] == '-'; ++i; }
    uint64_t v = 0;
    bool any = false, ovf = false;
    for (; i < len; ++i) {
        const uint32_t d = (uint32_t)p[i] - '0';
        if (d > 9) break;
        any = true;
        if (v > 0xFFFFFFFFFFFFFFFFull / 10 || (v == 0xFFFFFFFFFFFFFFFFull / 10 && d > 5)) ovf = true;
        else v = v * 10 + d;
    }
    if (!any) return 0;
    if (ovf) return 0xFFFFFFFFFFFFFFFFull;
    return neg ? (uint64_t)0 - v : v;
}

struct RangeSink {           // where per-column ranges of this wave accumulate
    uint64_t *mn, *mx;       // indexed by numeric index (LDS) or column (global)
    uint8_t *seen;           // indexed by column
    bool lds;
};

__device__ __forceinline__ void range_update(const RangeSink &r, uint32_t idx, uint64_t v)
{
    // unsigned 64-bit compare like the reference (ConvertToZDW.cpp:362-373,386-397)
    atomicMin(reinterpret_cast<unsigned long long *>(&r.mn[idx]), (unsigned long long)v);
    atomicMax(reinterpret_cast<unsigned long long *>(&r.mx[idx]), (unsigned long long)v);
}

__device__ inline void process_field(const Pass1Args &a, const RangeSink &rs, uint64_t fs, uint32_t len, uint32_t col, uint32_t row)
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
            id = dict_insert(a.in, a.extra, a.hi, a.slots, a.slot_len, a.slot_mask, a.max_entries, a.st, fs, len, a.trim);
        }
        if (in_block) *reinterpret_cast<uint32_t *>(recp) = id;
        return;
    }
    uint64_t v_range = 0, v_rec = 0;
    if (len) {
        if (kind == K_INT) {
            v_range = v_rec = parse_u64(p, len);
        } else {   // char(1): pass 1 adds byte 1 only after a backslash (:358-361), pass 2 always (:543-547)
            const int64_t c0 = (int8_t)p[0];
            const int64_t hi8 = (int64_t)((int)(int8_t)(len >= 2 ? p[1] : 0) * 256);
            v_range = (uint64_t)(p[0] == '\\' ? c0 + hi8 : c0);
            v_rec = (uint64_t)(c0 + hi8);
        }
        if (v_range > 0) range_update(rs, rs.lds ? (uint32_t)a.col_numidx[col] : col, v_range);
    }
    if (in_block) *reinterpret_cast<uint64_t *>(recp) = v_rec;
}

// ring entry: x = start[31:0], y = start[39:32] | col << 8, z = length, w = row
__device__ __forceinline__ void consume(const Pass1Args &a, const RangeSink &rs, const uint4 *ring, uint32_t head, uint32_t count,
                                        uint32_t lane)
{
    if (lane < count) {
        const uint4 e = ring[(head + lane) & (QCAP - 1)];
        const uint32_t col = e.y >> 8;
        if (col != COL_SKIP) process_field(a, rs, ((uint64_t)(e.y & 0xffu) << 32) | e.x, e.z, col, e.w);
    }
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

    uint32_t q_head = 0, q_tail = 0;   // wave-uniform, free running

    for (uint32_t si = blockIdx.x; si < a.nspans; si += gridDim.x) {
        const uint32_t rowbase = a.span_rowbase[si];
        if (a.span_rowbase[si + 1] == rowbase) continue;          // no row starts in this span
        if (rowbase > a.row_limit) break;                         // later spans hold rows of later blocks only
        if (__hip_atomic_load(&a.st->overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const uint64_t s0 = (a.span0 + si) * (uint64_t)SPAN_BYTES;
        const uint64_t ss = s0 > a.lo ? s0 : a.lo;
        const uint64_t se = s0 + SPAN_BYTES < a.hi ? s0 + SPAN_BYTES : a.hi;

        // wave-uniform running state of the open row / field
        const bool at_row_start = (ss == a.lo) || is_row_terminator(a.in, a.lo, ss - 1);
        uint64_t st_row_start = ss, st_field_start = ss;
        uint32_t st_tabs = 0, st_rows = 0;
        bool st_ignore = !at_row_start;      // the row open at ss began in an earlier span
        bool prev_term = at_row_start;
        uint64_t pos = ss & ~(uint64_t)(ROUND_BYTES - 1);
        bool bs_carry = pos > a.lo && a.in[pos - 1] == '\\';
        bool first = true, done = false;

        for (; !done && pos < a.hi; pos += ROUND_BYTES) {
            const uint32_t w = load_dword_guarded(a.in, pos + 4ull * lane, a.n);
            RoundMasks m;
            classify_round(a.in, a.lo, ss, a.hi, pos, w, lane, bs_carry, m);
            uint64_t term[4], tabm[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { term[k] = m.nl[k]; tabm[k] = m.tab[k]; }

            // the visit ends with the first terminator at or after se - 1
            uint64_t keep[4] = {~0ull, ~0ull, ~0ull, ~0ull};
            if (pos + ROUND_BYTES > se - 1) {
                const uint32_t thr = se - 1 > pos ? (uint32_t)(se - 1 - pos) : 0u;
                int cut = 1 << 30;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int l_lo = thr > (uint32_t)k ? (int)((thr - k + 3) >> 2) : 0;
                    const int lb = lowest_bit(term[k] & ~mask_lt(l_lo));
                    if (lb >= 0 && 4 * lb + k < cut) cut = 4 * lb + k;
                }
                if (cut < (1 << 30)) {
                    done = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        keep[k] = cut >= k ? mask_le((cut - k) >> 2) : 0ull;
                        term[k] &= keep[k];
                        tabm[k] &= keep[k];
                    }
                }
            }

            uint64_t pt[4];
            pt[0] = term[3] << 1; pt[1] = term[0]; pt[2] = term[1]; pt[3] = term[2];
            if (first) {
                const uint32_t fv = (uint32_t)(ss - pos);
                if (prev_term) pt[fv & 3] |= 1ull << (fv >> 2);
                first = false;
            } else if (prev_term) pt[0] |= 1ull;

            uint64_t sep[4], bnd[4], start[4];
            const uint64_t rel_lo = ss > pos ? ss - pos : 0;
            const uint64_t rel_hi = a.hi - pos < ROUND_BYTES ? a.hi - pos : ROUND_BYTES;
            uint32_t nsep_before = 0, nstart_before = 0, ntab_before = 0, nsep_total = 0;
            int tpos = -1, bpos = -1;
            const uint64_t lt = mask_lt((int)lane);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int l_lo = rel_lo > (uint64_t)k ? (int)((rel_lo - k + 3) >> 2) : 0;
                const int l_hi = rel_hi > (uint64_t)k ? (int)((rel_hi - k + 3) >> 2) : 0;
                const uint64_t valid = mask_lt(l_hi) & ~mask_lt(l_lo);
                const uint64_t blank = term[k] & pt[k];            // empty line: terminator right after a terminator
                bnd[k] = tabm[k] | term[k];
                sep[k] = bnd[k] & ~blank;
                start[k] = pt[k] & ~term[k] & valid & keep[k];
                nsep_before += mbcnt64(sep[k]);
                nstart_before += mbcnt64(start[k]);
                ntab_before += mbcnt64(tabm[k]);
                nsep_total += popc64(sep[k]);
                const int ht = highest_bit(term[k] & lt);
                if (ht >= 0 && 4 * ht + k > tpos) tpos = 4 * ht + k;
                const int hb = highest_bit(bnd[k] & lt);
                if (hb >= 0 && 4 * hb + k > bpos) bpos = 4 * hb + k;
            }

            // this lane's view of the open row / field at its first byte
            uint64_t my_row_start, my_field_start;
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
            my_field_start = bpos >= 0 ? pos + (uint64_t)bpos + 1 : st_field_start;
            uint32_t my_rows = st_rows + nstart_before;
            uint32_t qi = q_tail + nsep_before;

#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const uint64_t p = pos + 4ull * lane + k;
                if ((start[k] >> lane) & 1ull) ++my_rows;
                const bool is_term = (term[k] >> lane) & 1ull;
                const bool is_sep = (sep[k] >> lane) & 1ull;
                if (is_sep) {
                    const uint32_t row = rowbase + my_rows - 1;
                    uint32_t col = my_tabs;
                    if (my_ignore || col >= a.ncols || col >= COL_SKIP) col = COL_SKIP;
                    uint4 e;
                    e.x = (uint32_t)my_field_start;
                    e.y = (uint32_t)((my_field_start >> 32) & 0xffu) | (col << 8);
                    e.z = (uint32_t)(p - my_field_start);
                    e.w = row;
                    ring[qi & (QCAP - 1)] = e;
                    ++qi;
                    if (is_term && !my_ignore && row <= a.row_limit) {
                        if (my_tabs + 1 != a.ncols) atomicMin(&a.st->bad_row, row);       // ConvertToZDW.cpp:336-337
                        const uint64_t rl = p - my_row_start + 1;
                        if (rl >= FIRST_ROW_CAP - 1) {                                     // may grow m_LongestLine
                            const uint32_t li = atomicAdd(&a.st->long_count, 1u);
                            if (li < a.long_cap) { a.long_rows[li].start = my_row_start; a.long_rows[li].len = rl; }
                        }
                        if (row == a.row_limit) a.st->split_pos = my_row_start;
                    }
                }
                if (is_term) {
                    my_tabs = 0; my_row_start = p + 1; my_field_start = p + 1; my_ignore = false;
                } else if ((tabm[k] >> lane) & 1ull) {
                    ++my_tabs; my_field_start = p + 1;
                }
            }

            // lane 63 now holds the state at the end of the round
            st_row_start = bcast64(my_row_start, 63);
            st_field_start = bcast64(my_field_start, 63);
            st_tabs = bcast32(my_tabs, 63);
            st_rows = bcast32(my_rows, 63);
            st_ignore = bcast32(my_ignore ? 1u : 0u, 63) != 0;
            prev_term = (term[3] >> 63) & 1ull;
            q_tail += nsep_total;

            __syncthreads();
            while (q_tail - q_head >= 64) {
                consume(a, rs, ring, q_head, 64, lane);
                q_head += 64;
            }
            __syncthreads();
        }
    }
    __syncthreads();
    if (q_tail != q_head) consume(a, rs, ring, q_head, q_tail - q_head, lane);
    __syncthreads();

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
