// zdw_dev.h — device-side helpers shared by the gfx950 kernels of libzdw_hip.so.
//
// Everything here assumes wave64 (CDNA4) and is compiled for gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zdw {

constexpr uint32_t WAVE = 64;
constexpr uint32_t ROUND_BYTES = 256;   // bytes classified per wave step: one dword per lane
constexpr uint32_t SPAN_SHIFT_MIN = 14, SPAN_SHIFT_MAX = 18;   // a span = 2^shift input bytes whose row starts one wave visit owns (16-256 KiB, by block size)
constexpr uint32_t FIRST_ROW_CAP = 16384;

// column type ids (cplusplus/zdw_column_type_constants.h:17-35)
enum : uint8_t {
    T_VARCHAR = 0, T_TEXT = 1, T_DATETIME = 2, T_CHAR_2 = 3, T_CHAR = 6,
    T_TINY = 7, T_SHORT = 8, T_LONG = 9, T_LONGLONG = 10, T_DECIMAL = 11,
    T_TINY_SIGNED = 12, T_SHORT_SIGNED = 13, T_LONG_SIGNED = 14, T_LONGLONG_SIGNED = 15,
    T_TINYTEXT = 16, T_MEDIUMTEXT = 17, T_LONGTEXT = 18
};
// how pass 1 treats a column
enum : uint8_t { K_TEXT = 0, K_INT = 1, K_CHAR = 2 };

__host__ __device__ inline uint8_t kind_of_type(uint8_t t)
{
    switch (t) {
    case T_VARCHAR: case T_TEXT: case T_TINYTEXT: case T_MEDIUMTEXT: case T_LONGTEXT:
    case T_DATETIME: case T_CHAR_2: case T_DECIMAL: return K_TEXT;
    case T_CHAR: return K_CHAR;
    default: return K_INT;
    }
}

__device__ __forceinline__ uint32_t lane_id()
{
    return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of m in lanes below the calling lane
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// bits [0, b] set; b in [-1, 63]
__device__ __forceinline__ uint64_t mask_le(int b) { return b < 0 ? 0ull : ((2ull << b) - 1ull); }
// bits [0, b) set; b in [0, 64]
__device__ __forceinline__ uint64_t mask_lt(int b) { return b <= 0 ? 0ull : (b >= 64 ? ~0ull : ((1ull << b) - 1ull)); }
__device__ __forceinline__ int highest_bit(uint64_t m) { return m ? 63 - __clzll((long long)m) : -1; }
__device__ __forceinline__ int lowest_bit(uint64_t m) { return m ? __ffsll((long long)m) - 1 : -1; }
__device__ __forceinline__ uint32_t popc64(uint64_t m) { return (uint32_t)__popcll(m); }
// Lane masks.  BAL(one comparison) is a v_cmp that writes its scalar register pair; a mask built from several of them on the scalar
// unit costs the vector ALU nothing more, whereas the ballot of a bool that was combined or changed in a branch first goes through a
// vector register as 0 / 1 (v_cndmask + v_cmp).  lane_in(m): is the calling lane's bit set -- as a branch condition it is
// s_and_saveexec on m itself.  m must be wave-uniform.
#define BAL(cmp) __builtin_amdgcn_ballot_w64(cmp)
__device__ __forceinline__ bool lane_in(uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// m != 0, evaluated HERE: the compiler otherwise computes the comparison once and carries the answer from block to block -- which, for
// a wave-uniform bool, it does in a vector register (v_cndmask to set, v_cmp to test).  The empty asm makes m a new value each time.
__device__ __forceinline__ bool any_lane(uint64_t m) { asm volatile("" : "+s"(m)); return m != 0; }

__device__ __forceinline__ uint32_t bcast32(uint32_t v, int src_lane) { return (uint32_t)__shfl((int)v, src_lane, 64); }
__device__ __forceinline__ uint64_t bcast64(uint64_t v, int src_lane)
{
    uint32_t lo = bcast32((uint32_t)v, src_lane), hi = bcast32((uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
// ---- cross-lane data movement on the vector ALU (DPP): one v_*_dpp per step instead of an LDS-crossbar
// ---- permute; the scans below are dependent chains, so the latency of each step is what counts
// value of `v` in the lane selected by a DPP control; lanes without a source (or masked rows) get 0
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false); }
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHR1 = 0x138;
// inclusive prefix sum over the 64 lanes: Kogge-Stone inside each row of 16, then row totals forwarded
__device__ __forceinline__ uint32_t wave_incl_sum(uint32_t v)
{
    v += dpp_or_zero<DPP_ROW_SHR1, 0xf>(v);
    v += dpp_or_zero<DPP_ROW_SHR2, 0xf>(v);
    v += dpp_or_zero<DPP_ROW_SHR4, 0xf>(v);
    v += dpp_or_zero<DPP_ROW_SHR8, 0xf>(v);
    v += dpp_or_zero<DPP_ROW_BCAST15, 0xa>(v);      // lane 15 of rows 0 / 2 into rows 1 / 3
    v += dpp_or_zero<DPP_ROW_BCAST31, 0xc>(v);      // lane 31 into rows 2 and 3
    return v;
}
// exclusive prefix sum over the wave; total = the wave's sum
__device__ __forceinline__ uint32_t wave_excl_sum(uint32_t v, uint32_t lane, uint32_t &total)
{
    (void)lane;
    const uint32_t inc = wave_incl_sum(v);
    total = (uint32_t)__builtin_amdgcn_readlane((int)inc, 63);
    return inc - v;
}
// the value held by lane - 1 (0 in lane 0)
__device__ __forceinline__ uint32_t wave_shr1(uint32_t v) { return dpp_or_zero<DPP_WAVE_SHR1, 0xf>(v); }
// wave-uniform read of one lane's value (v_readlane_b32; the lane index must be wave-uniform)
__device__ __forceinline__ uint32_t lane_value(uint32_t v, int lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, lane); }

__device__ __forceinline__ uint32_t uniform32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform64(uint64_t v)
{
    return ((uint64_t)uniform32((uint32_t)(v >> 32)) << 32) | uniform32((uint32_t)v);
}

// ------------------------------------------------------------------------------------------
// Byte access to the TSV image.  `n` is the size of the allocation: nothing at or past it is read.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t load_dword_guarded(const uint8_t *in, uint64_t p, uint64_t n)
{
    if (p + 4 <= n) return *reinterpret_cast<const uint32_t *>(in + p);
    uint32_t w = 0;
    for (int k = 0; k < 4; ++k)
        if (p + k < n) w |= (uint32_t)in[p + k] << (8 * k);
    return w;
}

// One round's dword for this lane, branch-free.  Lanes whose dword would start at or past the end of
// the allocation read the last dword that starts inside it instead (an aligned dword whose first byte
// is inside an allocation never leaves its page); their bytes lie past `hi` and are masked by callers.
__device__ __forceinline__ uint32_t load_round_dword(const uint8_t *in, uint64_t pos, uint32_t lane, uint64_t n)
{
    const uint64_t last = (n - 1) & ~3ull;                    // n > 0
    const uint64_t base = pos <= last ? pos : last;            // wave-uniform
    const uint64_t room = last - base;
    const uint32_t maxoff = room < ROUND_BYTES - 4 ? (uint32_t)room : ROUND_BYTES - 4;
    const uint32_t off = 4u * lane < maxoff ? 4u * lane : maxoff;
    return *reinterpret_cast<const uint32_t *>(in + base + off);
}

// number of consecutive backslashes directly before position p, never looking below lo
__device__ inline uint32_t backslashes_before(const uint8_t *in, uint64_t lo, uint64_t p)
{
    uint32_t k = 0;
    while (p > lo + k && in[p - 1 - k] == '\\') ++k;
    return k;
}
// is byte p an unescaped newline (a row terminator)?  (getnextrow.cpp:45-53)
__device__ inline bool is_row_terminator(const uint8_t *in, uint64_t lo, uint64_t p)
{
    return in[p] == '\n' && (backslashes_before(in, lo, p) & 1u) == 0;
}

// 64-bit FNV-style hash over a byte range, finished with an avalanche; only used inside the
// dictionary table (never leaves the device).
__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}

}  // namespace zdw
