// zdw_field_dev.h — what pass 1 does with ONE field once its extent, column and row are known:
// dictionary insert for text-like columns, C-locale strtoull for integers, the two char(1) values,
// per-column ranges, and the fixed-width record pass 2 reads (cplusplus/ConvertToZDW.cpp:340-402).
//
// Two forms of the same work: the register form pass 1 runs on fields it can serve from its LDS
// byte ring (short plain-digit integers, strings of <= 24 bytes: see k_pass1.hip), and the general
// form below, which reads the field from global memory and takes everything else (-t trimming,
// signs / blanks / junk in numbers, long strings, contended table slots).
#pragma once
#include "zdw_dev.h"
#include "zdw_kernels.h"
#include "zdw_dict_dev.h"

namespace zdw {

// The dynamic LDS of the launching kernel
extern __shared__ __attribute__((aligned(16))) uint8_t zdw_lds[];

// ------------------------------------------------------------------------------------------
// integers
// ------------------------------------------------------------------------------------------
// C-locale strtoull(s, NULL, 10) on a byte range in global memory (ConvertToZDW.cpp:385,564)
__device__ inline uint64_t parse_u64_general(const uint8_t *p, uint32_t len)
{
    uint64_t v = 0;
    bool neg = false, any = false, ovf = false;
    uint32_t state = 0;     // 0 leading space, 1 digits
    for (uint32_t i = 0; i < len; ++i) {
        const uint32_t b = p[i];
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

// four ASCII digits, most significant in the low byte, already reduced by 0x30303030 -> value: two byte-wise dot products
// (v_dot4_u32_u8: weights 10, 1 on bytes 0, 1, then on bytes 2, 3 with 100 x the first sum as the addend) and one 24-bit multiply
__device__ __forceinline__ uint32_t four_digits(uint32_t t)
{
    const uint32_t hi = __builtin_amdgcn_udot4(t, 0x0000010au, 0u, false);          // 10 d0 + d1
    return __builtin_amdgcn_udot4(t, 0x010a0000u, __umul24(hi, 100u), false);       // + 10 d2 + d3
}
// bit 7 of a byte set where the byte of r is not an ASCII digit (t = r - 0x30303030); exact for the lowest bad byte
__device__ __forceinline__ uint32_t nondigit_bits(uint32_t r, uint32_t t) { return ((r + 0x46464646u) | t) & 0x80808080u; }

// ------------------------------------------------------------------------------------------
// column facts and ranges
// ------------------------------------------------------------------------------------------
// Per-workgroup tables in LDS (offsets into zdw_lds) and where column facts live.
struct ColTables {
    uint32_t meta;        // u32 per column: kind | numeric index << 2 | record offset << 16   (cols_in_lds)
    uint32_t mn, mx;      // u64 per numeric column                                              (cols_in_lds)
    uint32_t seen;        // u8 per column                                                       (cols_in_lds)
    bool cols_in_lds;     // else: column facts and ranges are read / updated in global memory
};

__device__ __forceinline__ uint32_t pack_col_meta(uint32_t kind, uint32_t numidx, uint32_t recoff)
{
    return kind | (numidx << 2) | (recoff << 16);
}

// unsigned 64-bit min / max like the reference (ConvertToZDW.cpp:362-373,386-397).  The plain reads only
// filter: min only falls and max only rises, so a stale value can cost a needless atomic, never lose an update.
__device__ __forceinline__ void range_update(const Pass1Args &a, const ColTables &t, uint32_t col, uint32_t numidx, uint64_t v)
{
    if (t.cols_in_lds) {
        unsigned long long *mn = reinterpret_cast<unsigned long long *>(zdw_lds + t.mn) + numidx;
        unsigned long long *mx = reinterpret_cast<unsigned long long *>(zdw_lds + t.mx) + numidx;
        if (v < *mn) atomicMin(mn, (unsigned long long)v);
        if (v > *mx) atomicMax(mx, (unsigned long long)v);
    } else {
        if (v < a.col_min[col]) atomicMin(reinterpret_cast<unsigned long long *>(&a.col_min[col]), (unsigned long long)v);
        if (v > a.col_max[col]) atomicMax(reinterpret_cast<unsigned long long *>(&a.col_max[col]), (unsigned long long)v);
    }
}
__device__ __forceinline__ void mark_seen(const Pass1Args &a, const ColTables &t, uint32_t col)
{
    if (t.cols_in_lds) { if (!zdw_lds[t.seen + col]) zdw_lds[t.seen + col] = 1; }
    else if (!a.col_seen[col]) a.col_seen[col] = 1;
}

// ------------------------------------------------------------------------------------------
// one field, general form (field bytes in global memory at a.in + fs)
// ------------------------------------------------------------------------------------------
template <bool BULK = false>
__device__ inline void process_field_general(const Pass1Args &a, const ColTables &t, const DictTable &d, IdAlloc &al, uint64_t fs, uint32_t len,
                                             uint32_t col, uint32_t row, bool in_block, bool &is_new)
{
    const uint8_t *p = a.in + fs;
    if (a.trim) while (len > 0 && p[len - 1] == ' ') --len;          // ConvertToZDW.cpp:295-313
    // column facts: packed in LDS (14-bit numeric index, 16-bit record offset) or, for schemas too wide for that, from global memory
    uint32_t kind, numidx, recoff;
    if (t.cols_in_lds) {
        const uint32_t meta = *reinterpret_cast<const uint32_t *>(zdw_lds + t.meta + 4 * col);
        kind = meta & 3u; numidx = (meta >> 2) & 0x3fffu; recoff = meta >> 16;
    } else {
        kind = a.col_kind[col]; numidx = (uint32_t)a.col_numidx[col]; recoff = a.col_recoff[col];
    }
    uint8_t *recp = a.rec + (uint64_t)row * a.rec_stride + recoff;
    if (kind == K_TEXT) {
        if (len) mark_seen(a, t, col);                                // empty fields are skipped (:340-341)
    }
    // the table insert is wave-level (ids are handed out by ballot): all calling lanes pass through it
    const bool want = kind == K_TEXT && len;
    if (__ballot(want)) {
        uint32_t id = 0;
        const uint64_t wm = __ballot(want);
        if (want) id = dict_insert<BULK>(p, a.in + a.n, a.in, a.n, a.extra, a.hi, d, al, ID_CHUNK, fs, len, a.trim != 0, is_new);
        adopt_ids(al, wm);
        if (want && in_block) {
            *reinterpret_cast<uint32_t *>(recp) = id;
            if (a.nz && id) atomicOr(a.nz + (uint64_t)row * a.nz_stride + (col >> 5), 1u << (col & 31u));
        }
    }
    if (kind == K_TEXT) {
        if (!want && in_block) *reinterpret_cast<uint32_t *>(recp) = 0u;
        return;
    }
    uint64_t v_range = 0, v_rec = 0;
    if (len) {
        if (kind == K_INT) {
            v_range = v_rec = parse_u64_general(p, len);
        } else {   // char(1): pass 1 adds byte 1 only after a backslash (:358-361), pass 2 always (:543-547)
            const uint32_t b0 = p[0];
            const int64_t c0 = (int8_t)b0;
            const int64_t hi8 = (int64_t)((int)(int8_t)(len >= 2 ? p[1] : 0) * 256);
            v_range = (uint64_t)(b0 == '\\' ? c0 + hi8 : c0);
            v_rec = (uint64_t)(c0 + hi8);
        }
        if (v_range > 0) range_update(a, t, col, numidx, v_range);
    }
    if (in_block) {
        *reinterpret_cast<uint64_t *>(recp) = v_rec;
        if (a.nz && v_rec) atomicOr(a.nz + (uint64_t)row * a.nz_stride + (col >> 5), 1u << (col & 31u));
    }
}

}  // namespace zdw
