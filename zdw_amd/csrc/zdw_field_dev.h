// zdw_field_dev.h — what pass 1 does with ONE field once its extent, column and row are known:
// dictionary insert for text-like columns, C-locale strtoull for integers, the two char(1) values,
// per-column ranges, and the fixed-width record pass 2 reads (cplusplus/ConvertToZDW.cpp:340-402).
#pragma once
#include "zdw_dev.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"

namespace zdw {

// ------------------------------------------------------------------------------------------
// field-level work
// ------------------------------------------------------------------------------------------
// C-locale strtoull(s, NULL, 10) on a byte range (ConvertToZDW.cpp:385,564): the
// general form, one 8-byte window load per 8 characters.
__device__ inline uint64_t parse_u64_general(const FieldBytes &f, uint32_t len)
{
    uint64_t win = 0, v = 0;
    bool neg = false, any = false, ovf = false;
    uint32_t state = 0;     // 0 leading space, 1 digits
    for (uint32_t i = 0; i < len; ++i) {
        if ((i & 7u) == 0) win = f.word(i, len - i < 8 ? len - i : 8u);
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
__device__ inline uint64_t parse_u64(const FieldBytes &f, uint32_t len)
{
    if (len <= 8) {
        const uint64_t w = pad_digits(f.word(0, len), len);
        if (all_digits(w)) return eight_digits(w);
    } else if (len <= 16) {
        const uint64_t w1 = pad_digits(f.word(0, len - 8), len - 8);
        const uint64_t w2 = f.word(len - 8, 8);
        if (all_digits(w1) && all_digits(w2)) return (uint64_t)eight_digits(w1) * 100000000ull + eight_digits(w2);
    }
    return parse_u64_general(f, len);
}

// Per-wave working set in LDS (offsets into zdw_lds) and where column facts live.
struct WaveCtx {
    uint32_t ring;        // descriptor ring
    uint32_t bytes;       // byte ring (BYTE_RING + 8)
    uint32_t meta;        // u32 per column: kind | numeric index << 2 | record offset << 16   (cols_in_lds)
    uint32_t mn, mx;      // u64 per numeric column                                              (cols_in_lds)
    uint32_t seen;        // u8 per column                                                       (cols_in_lds)
    bool cols_in_lds;     // else: column facts and ranges are read / updated in global memory
    uint64_t lds_lo, lds_hi;   // absolute positions currently held by the byte ring
};

__device__ __forceinline__ uint32_t pack_col_meta(uint32_t kind, uint32_t numidx, uint32_t recoff)
{
    return kind | (numidx << 2) | (recoff << 16);
}

// unsigned 64-bit min / max like the reference (ConvertToZDW.cpp:362-373,386-397).  The plain reads only
// filter: min only falls and max only rises, so a stale value can cost a needless atomic, never lose an update.
__device__ __forceinline__ void range_update(const Pass1Args &a, const WaveCtx &w, uint32_t col, uint32_t numidx, uint64_t v)
{
    if (w.cols_in_lds) {
        unsigned long long *mn = reinterpret_cast<unsigned long long *>(zdw_lds + w.mn) + numidx;
        unsigned long long *mx = reinterpret_cast<unsigned long long *>(zdw_lds + w.mx) + numidx;
        if (v < *mn) atomicMin(mn, (unsigned long long)v);
        if (v > *mx) atomicMax(mx, (unsigned long long)v);
    } else {
        if (v < a.col_min[col]) atomicMin(reinterpret_cast<unsigned long long *>(&a.col_min[col]), (unsigned long long)v);
        if (v > a.col_max[col]) atomicMax(reinterpret_cast<unsigned long long *>(&a.col_max[col]), (unsigned long long)v);
    }
}

__device__ inline void process_field(const Pass1Args &a, const WaveCtx &w, uint64_t fs, uint32_t len, uint32_t col, uint32_t row,
                                     bool &is_new)
{
    const bool in_block = row < a.row_limit;
    if (!in_block && !(row == a.row_limit && col < a.partial_fields)) return;
    FieldBytes f;
    f.g = a.in + fs; f.g_end = a.in + a.n; f.ring = w.bytes; f.at = (uint32_t)fs & (BYTE_RING - 1);
    f.in_lds = fs >= w.lds_lo && fs + len <= w.lds_hi;
    if (a.trim) while (len > 0 && f.byte(len - 1) == ' ') --len;          // ConvertToZDW.cpp:295-313
    const uint32_t meta = w.cols_in_lds ? *reinterpret_cast<const uint32_t *>(zdw_lds + w.meta + 4 * col)
                                        : pack_col_meta(a.col_kind[col], (uint32_t)a.col_numidx[col] & 0x3fffu, a.col_recoff[col]);
    const uint32_t kind = meta & 3u;
    uint8_t *recp = a.rec + (uint64_t)row * a.rec_stride + (meta >> 16);
    if (kind == K_TEXT) {
        uint32_t id = 0;
        if (len) {                                                    // empty fields are skipped (:340-341)
            if (w.cols_in_lds) { if (!zdw_lds[w.seen + col]) zdw_lds[w.seen + col] = 1; }
            else if (!a.col_seen[col]) a.col_seen[col] = 1;
            if (!(a.debug_skip & 2u))
                id = dict_insert(f, a.in, a.n, a.extra, a.hi, a.slots, a.slot_len, a.slot_mask, a.st, fs, len, a.trim, is_new);
        }
        if (in_block && !(a.debug_skip & 8u)) *reinterpret_cast<uint32_t *>(recp) = id;
        return;
    }
    uint64_t v_range = 0, v_rec = 0;
    if (len) {
        if (kind == K_INT) {
            v_range = v_rec = (a.debug_skip & 4u) ? 1 : parse_u64(f, len);
        } else {   // char(1): pass 1 adds byte 1 only after a backslash (:358-361), pass 2 always (:543-547)
            const uint32_t b0 = f.byte(0);
            const int64_t c0 = (int8_t)b0;
            const int64_t hi8 = (int64_t)((int)(int8_t)(len >= 2 ? f.byte(1) : 0) * 256);
            v_range = (uint64_t)(b0 == '\\' ? c0 + hi8 : c0);
            v_rec = (uint64_t)(c0 + hi8);
        }
        if (v_range > 0) range_update(a, w, col, (meta >> 2) & 0x3fffu, v_range);
    }
    if (in_block && !(a.debug_skip & 8u)) *reinterpret_cast<uint64_t *>(recp) = v_rec;
}

}  // namespace zdw
