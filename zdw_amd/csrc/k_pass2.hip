// k_pass2.hip — pass 2 of the ZDW block encode on gfx950 (ConvertToZDW::writeBlockRows,
// cplusplus/ConvertToZDW.cpp:486-606).
//
// Input is the fixed-width record matrix rec[row][column] that pass 1 left in HBM (u32 dictionary
// table slot id per text field, u64 parsed value per numeric field), so this pass is pure
// fixed-stride integer work: stored value = dictionary offset | value - base, flag = stored value
// differs from the previous row's, row bytes = ceil(used/8) flag bytes followed by the low
// `size` bytes of every flagged value in column order.
//
// Two launches: row lengths (then an exclusive scan gives each row's output offset), then emit.
//
// Column kernels (17 or more used columns; the size kernel from 8): a workgroup takes a chunk of rows, a wave a
// 64-column batch of it (lane = column; with fewer than three batches several waves share one and split the rows),
// so "the row above" is what the lane read one step earlier.  The chunk's records are streamed through LDS with
// coalesced 16-byte loads; the dictionary gathers of a chunk are in flight together; the emit kernel assembles the
// chunk's rows in LDS and writes them out with aligned 16-byte stores (the byte-granular scatter of the values
// stays inside the LDS) while the next chunk's records are on their way.
//
// Very narrow schemas: a row is handled by G = pow2 >= used columns adjacent lanes (64/G rows per wave):
// flag bits come from one wave ballot, the byte position of each value from a G-lane prefix sum.
#include "zdw_dev.h"
#include "zdw_knobs.h"
#include "zdw_kernels.h"

#include <algorithm>
#include <cstdlib>

namespace zdw {

struct __attribute__((packed)) PackedW64 { uint64_t v; };
struct __attribute__((packed)) PackedW32 { uint32_t v; };
struct __attribute__((packed)) PackedW16 { uint16_t v; };

// what pass 1 left for (row, used column u): unique id + 1 (text) or the parsed value (numeric)
__device__ __forceinline__ uint64_t raw_value(const Pass2Args &a, uint32_t row, uint32_t off, bool text)
{
    const uint8_t *p = a.rec + (uint64_t)row * a.rec_stride + off;
    return text ? (uint64_t)*reinterpret_cast<const uint32_t *>(p) : *reinterpret_cast<const uint64_t *>(p);
}
// the value the file stores for it: dictionary offset (ConvertToZDW.cpp:528-531) or value - base
// (:543-547, :564-566).  The mapping is injective per column (distinct strings have distinct offsets,
// v -> v - base is one-to-one, 0 <-> empty), so "differs from the previous row" can be decided on the
// raw values and the dictionary gather is only paid for values that are actually written.
__device__ __forceinline__ uint64_t stored_from_raw(const Pass2Args &a, uint64_t raw, uint32_t u, bool text)
{
    if (!raw) return 0ull;
    return text ? (uint64_t)a.id_off[(uint32_t)raw - 1] : raw - a.used_base[u];
}
// flag of the block's first row: its stored value against the row before the block (the halo row of a shard arrives as
// stored values; without one it is zero, and a text field then differs iff it is not empty: dictionary offsets are >= 1)
__device__ __forceinline__ bool first_row_flag(const Pass2Args &a, uint64_t raw, uint32_t u, bool text)
{
    if (a.prev_is_zero && text) return raw != 0;
    return stored_from_raw(a, raw, u, text) != a.prev0[u];
}
__device__ __forceinline__ uint64_t stored_value(const Pass2Args &a, uint32_t row, uint32_t u)
{
    const bool text = a.used_kind[u] == K_TEXT;
    return stored_from_raw(a, raw_value(a, row, a.used_recoff[u], text), u, text);
}

template <uint32_t G, bool EMIT>
__global__ __launch_bounds__(256) void k_pass2(const Pass2Args a)
{
    constexpr uint32_t ROWS_PER_WAVE = 64 / G;
    const uint32_t lane = lane_id();
    const uint32_t sub = lane & (G - 1), grp = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t row64 = wave * ROWS_PER_WAVE + grp;
    const bool row_ok = row64 < a.nrows;
    const uint32_t row = (uint32_t)row64;
    const uint32_t nflag = (a.nused + 7) >> 3;
    const uint64_t out_row = (EMIT && row_ok) ? a.row_off[row] : 0ull;
    uint32_t running = 0;      // value bytes of this row emitted so far (uniform inside the group)

    for (uint32_t u0 = 0; u0 < a.nused; u0 += G) {
        const uint32_t u = u0 + sub;
        const bool ok = row_ok && u < a.nused;
        uint64_t cur = 0;
        uint32_t sz = 0;
        bool flag = false, text = false;
        if (ok) {
            text = a.used_kind[u] == K_TEXT;
            const uint32_t off = a.used_recoff[u];
            cur = raw_value(a, row, off, text);
            sz = a.used_size[u];
            if (row) flag = cur != raw_value(a, row - 1, off, text);
            else flag = first_row_flag(a, cur, u, text);
        }
        const uint32_t mine = flag ? sz : 0u;
        // inclusive prefix over the G lanes of the group
        uint32_t inc = mine;
#pragma unroll
        for (uint32_t d = 1; d < G; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, d, G);
            if (sub >= d) inc += t;
        }
        const uint32_t tot = (uint32_t)__shfl((int)inc, G - 1, G);
        if (EMIT) {
            const uint64_t bal = __ballot(flag);
            uint64_t bits = bal;
            if constexpr (G < 64) bits = (bal >> (grp * G)) & ((1ull << G) - 1ull);
            constexpr uint32_t FB = G >= 8 ? G / 8 : 1;                  // flag bytes this chunk can produce
            if (row_ok && sub < FB && u0 + 8 * sub < a.nused)
                a.out[out_row + (u0 >> 3) + sub] = (uint8_t)(bits >> (8 * sub));
            if (flag) {
                const uint64_t v = stored_from_raw(a, cur, u, text);
                // little-endian truncating store in at most three pieces (4 + 2 + 1 bytes; gfx950 stores unaligned)
                uint8_t *q = a.out + out_row + nflag + running + (inc - mine);
                if (sz == 8) { reinterpret_cast<PackedW64 *>(q)->v = v; }
                else {
                    uint64_t r = v;
                    if (sz & 4u) { reinterpret_cast<PackedW32 *>(q)->v = (uint32_t)r; q += 4; r >>= 32; }
                    if (sz & 2u) { reinterpret_cast<PackedW16 *>(q)->v = (uint16_t)r; q += 2; r >>= 16; }
                    if (sz & 1u) *q = (uint8_t)r;
                }
            }
        }
        running += tot;
    }
    if (!EMIT && row_ok && sub == 0) a.row_len[row] = nflag + running;
}

// Records stream through once and the rows leave once: non-temporal loads and stores, so that they do not push the dictionary offsets
// (id_off: the emit kernel's gather, a few KB of it hot) out of the CU's vector L1.
typedef uint32_t p2_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 stream_load16(const uint8_t *p)
{
    const p2_u32x4 x = __builtin_nontemporal_load(reinterpret_cast<const p2_u32x4 *>(p));
    return make_uint4(x.x, x.y, x.z, x.w);
}
__device__ __forceinline__ void stream_store16(uint8_t *p, const uint4 &v)
{
    p2_u32x4 x;
    x.x = v.x; x.y = v.y; x.z = v.z; x.w = v.w;
    __builtin_nontemporal_store(x, reinterpret_cast<p2_u32x4 *>(p));
}

// The same for schemas of up to G used columns (one column per lane: all of them but those of more than 64 columns), with a loop: a
// wave takes P2N_ITERS groups of 64 / G consecutive rows.  The kernel above starts a wave per 64 / G rows -- 2.1 M workgroups for the
// 67 M rows of the C2 shape, each with three dependent round trips (the column's facts, the records, the row's offset) and nothing to
// cover them: 1 TB/s.  Here the column's facts are loaded once per wave, and the records and the offset of the next group are
// requested before the current one is worked on.
// STAGE (emit): the wave's rows -- consecutive, so their bytes are one piece of the output -- are put together in LDS (at the byte
// phase of their place in the output) and leave as aligned 16-byte stores, instead of every lane storing its 1 to 8 bytes on its own.
constexpr uint32_t P2N_ITERS = 16;
__host__ __device__ inline uint32_t p2n_image_bytes(uint32_t g, uint32_t out_row_max) { return ((64u / g) * P2N_ITERS * out_row_max + 32u + 15u) & ~15u; }
template <uint32_t G, bool EMIT, bool STAGE = false>
__global__ __launch_bounds__(256) void k_pass2_loop(const Pass2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t p2n_lds[];
    constexpr uint32_t ROWS_PER_WAVE = 64 / G;
    const uint32_t lane = lane_id();
    const uint32_t sub = lane & (G - 1), grp = lane / G;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t nflag = (a.nused + 7) >> 3;
    const uint32_t u = sub;
    const bool col_ok = u < a.nused;
    bool text = false;
    uint32_t off = 0, sz = 0;
    if (col_ok) { text = a.used_kind[u] == K_TEXT; off = a.used_recoff[u]; sz = a.used_size[u]; }
    const uint64_t first = wave * (ROWS_PER_WAVE * P2N_ITERS) + grp;
    // (STAGE) the wave's piece of the output: [obase, oend), its image in LDS from byte `shift` on
    uint8_t *img = nullptr;
    uint64_t obase = 0;
    uint32_t shift = 0, onbytes = 0;
    if (EMIT && STAGE) {
        const uint64_t r0 = wave * (ROWS_PER_WAVE * P2N_ITERS);
        if (r0 >= a.nrows) return;
        const uint64_t r1 = r0 + ROWS_PER_WAVE * P2N_ITERS < a.nrows ? r0 + ROWS_PER_WAVE * P2N_ITERS : a.nrows;
        obase = a.row_off[r0];
        onbytes = (uint32_t)(a.row_off[r1] - obase);
        shift = (uint32_t)obase & 15u;
        img = p2n_lds + (threadIdx.x >> 6) * p2n_image_bytes(G, a.out_row_max);
    }
    // the group in flight: this row's record, the record of the row above, where the row's output begins
    uint64_t cur = 0, above = 0, out_row = 0;
    auto request = [&](uint64_t row64) {
        cur = above = 0; out_row = 0;
        if (row64 < a.nrows && col_ok) {
            cur = raw_value(a, (uint32_t)row64, off, text);
            if (row64) above = raw_value(a, (uint32_t)row64 - 1u, off, text);
        }
        if (EMIT && row64 < a.nrows) out_row = a.row_off[row64];
    };
    request(first);
    for (uint32_t it = 0; it < P2N_ITERS; ++it) {
        const uint64_t row64 = first + (uint64_t)it * ROWS_PER_WAVE;
        if (row64 - grp >= a.nrows) break;                      // (wave-uniform: the group's first row)
        const uint64_t c = cur, ab = above, orow = out_row;
        if (it + 1 < P2N_ITERS) request(row64 + ROWS_PER_WAVE);
        const bool ok = row64 < a.nrows && col_ok;
        bool flag = false;
        if (ok) flag = row64 ? c != ab : first_row_flag(a, c, u, text);
        const uint32_t mine = flag ? sz : 0u;
        uint32_t inc = mine;                                    // inclusive prefix over the G lanes of the group
#pragma unroll
        for (uint32_t d = 1; d < G; d <<= 1) {
            const uint32_t t = (uint32_t)__shfl_up((int)inc, d, G);
            if (sub >= d) inc += t;
        }
        const uint32_t tot = (uint32_t)__shfl((int)inc, G - 1, G);
        if (EMIT) {
            const uint64_t bal = __ballot(flag);
            uint64_t bits = bal;
            if constexpr (G < 64) bits = (bal >> (grp * G)) & ((1ull << G) - 1ull);
            constexpr uint32_t FB = G >= 8 ? G / 8 : 1;                  // flag bytes of the row
            uint8_t *rowp = STAGE ? img + shift + (uint32_t)(orow - obase) : a.out + orow;      // (LDS stores are unaligned-capable too)
            if (row64 < a.nrows && sub < FB && 8 * sub < a.nused) rowp[sub] = (uint8_t)(bits >> (8 * sub));
            if (flag) {
                const uint64_t v = stored_from_raw(a, c, u, text);
                uint8_t *q = rowp + nflag + (inc - mine);
                if (sz == 8) { reinterpret_cast<PackedW64 *>(q)->v = v; }
                else {
                    uint64_t r = v;
                    if (sz & 4u) { reinterpret_cast<PackedW32 *>(q)->v = (uint32_t)r; q += 4; r >>= 32; }
                    if (sz & 2u) { reinterpret_cast<PackedW16 *>(q)->v = (uint16_t)r; q += 2; r >>= 16; }
                    if (sz & 1u) *q = (uint8_t)r;
                }
            }
        } else if (row64 < a.nrows && sub == 0) a.row_len[row64] = nflag + tot;
    }
    if (EMIT && STAGE) {
        // the image leaves: whole 16-byte vectors where it covers them, bytes at its two ends
        __builtin_amdgcn_wave_barrier();                         // (one wave per image: its LDS operations execute in order)
        uint8_t *gdst = a.out + (obase - shift);                 // 16-byte aligned
        const uint32_t endb = shift + onbytes;
        for (uint32_t o = lane * 16u; o < endb; o += 64u * 16u) {
            if (o >= shift && o + 16u <= endb) stream_store16(gdst + o, *reinterpret_cast<const uint4 *>(img + o));
            else {
                const uint32_t lo = o > shift ? o : shift, hi = o + 16u < endb ? o + 16u : endb;
                for (uint32_t k = lo; k < hi; ++k) gdst[k] = img[k];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// column kernels: a wave owns 64 columns of (its share of) a chunk of rows
// ------------------------------------------------------------------------------------------
// Lane = column, so "the row above" is the value the lane loaded one step earlier: the record matrix is
// read once, and all loads of a chunk (P2_R + 1 per lane) are in flight together.
constexpr uint32_t P2_R_SIZE = 16;              // rows per chunk, size kernel
                                                // (emit kernel: 8 rows per wave, 4 for very wide rows, 16 on request: chosen at launch)
constexpr uint32_t P2_LDS_BUDGET = 40 * 1024;    // output image of one row chunk (emit)

struct ColFacts { uint32_t off; uint32_t sz; bool text; bool ok; };

__device__ __forceinline__ ColFacts col_facts(const Pass2Args &a, uint32_t u)
{
    ColFacts f;
    f.ok = u < a.nused;
    f.off = f.ok ? a.used_recoff[u] : 0u;
    f.sz = f.ok ? a.used_size[u] : 0u;
    f.text = f.ok && a.used_kind[u] == K_TEXT;
    return f;
}
// one 8-byte load whatever the column kind (a text id is the low half; the matrix has slack behind its last row)
__device__ __forceinline__ uint64_t raw_at(const uint8_t *rowp, const ColFacts &f)
{
    const uint64_t x = reinterpret_cast<const PackedW64 *>(rowp + f.off)->v;
    return f.text ? (x & 0xffffffffull) : x;
}
// raw values of column f for rows row0 - 1 .. row0 + rows - 1 in v[0 .. rows]; v[0] = 0 for the block's first row
template <uint32_t P2_R>
__device__ __forceinline__ void load_column(const Pass2Args &a, const ColFacts &f, uint32_t row0, uint32_t rows, uint64_t (&v)[P2_R + 1])
{
    const uint8_t *rp = a.rec + (uint64_t)row0 * a.rec_stride;
    v[0] = (f.ok && row0) ? raw_at(rp - a.rec_stride, f) : 0ull;
#pragma unroll
    for (uint32_t r = 0; r < P2_R; ++r) v[r + 1] = (f.ok && r < rows) ? raw_at(rp + (uint64_t)r * a.rec_stride, f) : 0ull;
}

// Fewer than three 64-column batches: the four waves of a workgroup share them, 4 or 2 waves per batch, each
// taking its own rows of a chunk that is as many times longer (no wave idles on schemas of 33..128 columns).
__host__ __device__ inline uint32_t p2_waves_per_batch(uint32_t nbatch) { return nbatch <= 1 ? 4u : (nbatch == 2 ? 2u : 1u); }

// ---- staged form: the workgroup streams the chunk's records (rows row0 - 1 .. row0 + rows - 1, one contiguous
// piece of the matrix) into LDS with coalesced 16-byte loads and the lanes pick their columns from there.  Lane =
// column loads straight from HBM reach half the rate of a streaming read (8 bytes per lane at a 4..8-byte pitch).
constexpr uint32_t P2_STAGE_BUDGET = 48 * 1024;   // staged when (P2_R + 1) rows of records fit in this much LDS

__host__ __device__ inline uint32_t p2_stage_bytes(uint32_t rows, uint32_t rec_stride) { return (rows + 1) * rec_stride + 32u; }

// returns the LDS offset (inside `stage`) of row row0's record; callers __syncthreads() before reading
__device__ __forceinline__ uint32_t stage_chunk(const Pass2Args &a, uint8_t *stage, uint32_t row0, uint32_t rows)
{
    const uint32_t first = row0 ? row0 - 1 : 0;
    const uint8_t *src = a.rec + (uint64_t)first * a.rec_stride;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
    const uint32_t nbytes = mis + (row0 + rows - first) * a.rec_stride;        // from the aligned start
    const uint8_t *src16 = src - mis;
    for (uint32_t o = threadIdx.x * 16u; o < nbytes; o += 256u * 16u)
        *reinterpret_cast<uint4 *>(stage + o) = stream_load16(src16 + o);     // may run up to 15 bytes past the piece: the matrix has slack
    return mis + (row0 - first) * a.rec_stride;
}
// a record field from the staged copy (fields sit at multiples of 4)
__device__ __forceinline__ uint64_t staged_at(const uint8_t *stage, uint32_t at, const ColFacts &f)
{
    const uint32_t *q = reinterpret_cast<const uint32_t *>(stage + at + f.off);
    const uint32_t lo = q[0], hi = q[1];
    return f.text ? (uint64_t)lo : (((uint64_t)hi << 32) | lo);
}
template <uint32_t P2_R>
__device__ __forceinline__ void load_column_staged(const uint8_t *stage, uint32_t at0, uint32_t stride, const ColFacts &f, uint32_t row0, uint32_t rows,
                                                   uint64_t (&v)[P2_R + 1])
{
    v[0] = (f.ok && row0) ? staged_at(stage, at0 - stride, f) : 0ull;
#pragma unroll
    for (uint32_t r = 0; r < P2_R; ++r) v[r + 1] = (f.ok && r < rows) ? staged_at(stage, at0 + r * stride, f) : 0ull;
}

// row lengths + for every (row, 64-column batch) the value bytes the batches before it take in that row
template <bool STAGED>
__global__ __launch_bounds__(256) void k_pass2_size_cols(const Pass2Args a)
{
    constexpr uint32_t P2_R = P2_R_SIZE;
    extern __shared__ __attribute__((aligned(16))) uint8_t p2s_lds[];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t nflag = (a.nused + 7) >> 3;
    const uint32_t nbatch = (a.nused + 63) >> 6;
    const uint32_t wpb = p2_waves_per_batch(nbatch), nbe = 4u / wpb;      // waves per batch; batches handled side by side
    const uint32_t CR = P2_R * wpb;                                          // rows per chunk
    uint16_t *tot = reinterpret_cast<uint16_t *>(p2s_lds);                  // [CR][nbatch]
    uint8_t *stage = p2s_lds + ((CR * nbatch * 2u + 15u) & ~15u);
    const uint32_t nchunks = (a.nrows + CR - 1) / CR;
    const uint32_t b0 = wv % nbe, sub = wv / nbe;
    const ColFacts f0 = col_facts(a, 64 * b0 + lane);          // the wave's first batch is the same in every chunk
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t crow0 = chunk * CR;
        const uint32_t crows = a.nrows - crow0 < CR ? a.nrows - crow0 : CR;
        uint32_t at0 = 0;
        if (STAGED) { at0 = stage_chunk(a, stage, crow0, crows); __syncthreads(); }
        // this wave's rows of the chunk
        const uint32_t r0 = sub * P2_R;
        const uint32_t row0 = crow0 + r0;
        const uint32_t rows = crows > r0 ? (crows - r0 < P2_R ? crows - r0 : P2_R) : 0u;
        for (uint32_t b = b0; b < nbatch && rows; b += nbe) {
            const uint32_t u = 64 * b + lane;
            const ColFacts f = b == b0 ? f0 : col_facts(a, u);
            uint64_t v[P2_R + 1];
            if (STAGED) load_column_staged<P2_R>(stage, at0 + r0 * a.rec_stride, a.rec_stride, f, row0, rows, v);
            else load_column<P2_R>(a, f, row0, rows, v);
#pragma unroll
            for (uint32_t r = 0; r < P2_R; ++r) {
                bool flag = v[r + 1] != v[r];
                if (row0 + r == 0 && f.ok) flag = first_row_flag(a, v[1], u, f.text);
                const uint32_t t = lane_value(wave_incl_sum(flag ? f.sz : 0u), 63);
                if (lane == 0) tot[(r0 + r) * nbatch + b] = (uint16_t)t;
            }
        }
        __syncthreads();
        if (threadIdx.x < crows) {
            const uint32_t r = threadIdx.x;
            uint32_t run = 0;
            for (uint32_t b = 0; b < nbatch; ++b) {
                const uint32_t t = tot[r * nbatch + b];
                a.batch_pre[(uint64_t)(crow0 + r) * nbatch + b] = (uint16_t)run;
                run += t;
            }
            a.row_len[crow0 + r] = nflag + run;
        }
        __syncthreads();
    }
}

struct __attribute__((packed)) PackedL64 { uint64_t v; };
struct __attribute__((packed)) PackedL32 { uint32_t v; };
struct __attribute__((packed)) PackedL16 { uint16_t v; };

// emit: the workgroup assembles P2_R consecutive rows in LDS (at the byte offset that makes LDS and global
// 16-byte phases agree) and copies the image out with aligned vector stores
// one wave, one 64-column batch of a chunk: flags, dictionary gathers, flag bytes and values into the LDS image
template <uint32_t P2_R>
__device__ __forceinline__ void emit_batch(const Pass2Args &a, uint8_t *img, uint32_t b, uint32_t lane, const ColFacts &f, uint64_t (&v)[P2_R + 1],
                                           uint32_t row0, uint32_t rows, uint32_t my_at, uint32_t my_pre, uint32_t nflag)
{
    // my_at / my_pre (lanes 0 .. rows - 1): where row `lane` begins in the image; the value bytes the batches before b take in it
    const uint32_t u = 64 * b + lane;
    // flags first, then every dictionary gather of the chunk in flight together
    uint32_t flags = 0;
#pragma unroll
    for (uint32_t r = 0; r < P2_R; ++r) flags |= (v[r + 1] != v[r] ? 1u : 0u) << r;
    if (row0 == 0 && f.ok) flags = (flags & ~1u) | (first_row_flag(a, v[1], u, f.text) ? 1u : 0u);
    if (!f.ok) flags = 0;
    const uint64_t basev = f.ok && !f.text ? a.used_base[u] : 0ull;
#pragma unroll
    for (uint32_t r = 0; r < P2_R; ++r) {
        uint64_t x = v[r + 1];
        if ((flags >> r) & 1u) x = f.text ? (x ? (uint64_t)a.id_off[(uint32_t)x - 1] : 0ull) : (x ? x - basev : 0ull);
        v[r + 1] = x;
    }
#pragma unroll
    for (uint32_t r = 0; r < P2_R; ++r) {
        if (r >= rows) continue;
        const bool flag = (flags >> r) & 1u;
        const uint32_t mine = flag ? f.sz : 0u;
        const uint32_t inc = wave_incl_sum(mine);
        const uint64_t bal = __ballot(flag);
        const uint32_t at = lane_value(my_at, r), pre = lane_value(my_pre, r);
        if (lane < 8 && 64 * b + 8 * lane < a.nused) img[at + 8 * b + lane] = (uint8_t)(bal >> (8 * lane));
        if (flag) {
            // little-endian truncating store in at most three pieces (4 + 2 + 1 bytes; LDS stores are unaligned-capable).  (Measured
            // and dropped: ORs into the aligned dwords of a zeroed image -- no unaligned LDS stalls any more, the kernel 1 % faster,
            // but 5 GB more fetched from beyond L2 per 10 GB: DESIGN.md 4.3.)
            uint8_t *q = img + at + nflag + pre + (inc - mine);
            const uint32_t sz = f.sz;
            uint64_t x = v[r + 1];
            if (sz == 8) { reinterpret_cast<PackedL64 *>(q)->v = x; }
            else {
                if (sz & 4u) { reinterpret_cast<PackedL32 *>(q)->v = (uint32_t)x; q += 4; x >>= 32; }
                if (sz & 2u) { reinterpret_cast<PackedL16 *>(q)->v = (uint16_t)x; q += 2; x >>= 16; }
                if (sz & 1u) *q = (uint8_t)x;
            }
        }
    }
}

// image [shift, shift + nbytes) -> out[base, base + nbytes): whole 16-byte vectors where they are covered
__device__ __forceinline__ void copy_image_out(const Pass2Args &a, const uint8_t *img, uint64_t base, uint32_t shift, uint32_t nbytes)
{
    uint8_t *gdst = a.out + (base - shift);              // 16-byte aligned
    const uint32_t endb = shift + nbytes;
    for (uint32_t o = threadIdx.x * 16u; o < endb; o += 256u * 16u) {
        if (o >= shift && o + 16u <= endb) {
            stream_store16(gdst + o, *reinterpret_cast<const uint4 *>(img + o));
        } else {
            const uint32_t lo = o > shift ? o : shift, hi = o + 16u < endb ? o + 16u : endb;
            for (uint32_t k = lo; k < hi; ++k) gdst[k] = img[k];
        }
    }
}

// emit: the workgroup assembles P2_R consecutive rows in LDS (at the byte offset that makes LDS and global
// 16-byte phases agree) and copies the image out with aligned vector stores.
// STAGED: the chunk's records come through LDS (stage_chunk) instead of lane = column loads from HBM.
template <bool STAGED, uint32_t P2_R>
__global__ __launch_bounds__(256) void k_pass2_emit_cols(const Pass2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t p2_lds[];
    uint8_t *stage = p2_lds + ((P2_R * a.out_row_max + 32u + 15u) & ~15u);
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t nflag = (a.nused + 7) >> 3;
    const uint32_t nbatch = (a.nused + 63) >> 6;
    const uint32_t nchunks = (a.nrows + P2_R - 1) / P2_R;
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t row0 = chunk * P2_R;
        const uint32_t rows = a.nrows - row0 < P2_R ? a.nrows - row0 : P2_R;
        const uint64_t base = a.row_off[row0];
        const uint32_t nbytes = (uint32_t)(a.row_off[row0 + rows] - base);
        const uint32_t shift = (uint32_t)base & 15u;
        uint32_t at0 = 0;
        if (STAGED) { at0 = stage_chunk(a, stage, row0, rows); __syncthreads(); }
        for (uint32_t b = wv; b < nbatch; b += 4) {
            const ColFacts f = col_facts(a, 64 * b + lane);
            uint64_t v[P2_R + 1];
            if (STAGED) load_column_staged<P2_R>(stage, at0, a.rec_stride, f, row0, rows, v);
            else load_column<P2_R>(a, f, row0, rows, v);
            uint32_t my_at = 0, my_pre = 0;
            if (lane < rows) {
                my_at = (uint32_t)(a.row_off[row0 + lane] - base) + shift;
                my_pre = a.batch_pre[(uint64_t)(row0 + lane) * nbatch + b];
            }
            emit_batch<P2_R>(a, p2_lds, b, lane, f, v, row0, rows, my_at, my_pre, nflag);
        }
        __syncthreads();
        copy_image_out(a, p2_lds, base, shift, nbytes);
        __syncthreads();
    }
}

// all LDS operations of the workgroup so far are done and visible; global stores and loads stay in flight
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The staged form for at most four batches (one per wave) and at most 16 KiB of records per chunk.  A wave holds
// its columns in registers right after the chunk starts, so the next chunk's records, row offsets and batch
// prefixes are requested then and arrive while this chunk's gathers and image writes run; they are committed to
// LDS between the image barrier and the copy-out.  The barriers order LDS only: the copy-out stores are never
// waited for inside the loop.
template <uint32_t P2_R>
__global__ __launch_bounds__(256, P2_R <= 8 ? 7 : 4) void k_pass2_emit_cols_piped(const Pass2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t p2_lds[];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t nflag = (a.nused + 7) >> 3;
    const uint32_t nbatch = (a.nused + 63) >> 6;
    const uint32_t wpb = p2_waves_per_batch(nbatch), nbe = 4u / wpb;      // waves per batch; batches side by side
    const uint32_t CR = P2_R * wpb;                                          // rows per chunk
    uint8_t *stage = p2_lds + ((CR * a.out_row_max + 32u + 15u) & ~15u);
    const uint32_t nchunks = (a.nrows + CR - 1) / CR;
    if (blockIdx.x >= nchunks) return;
    const uint32_t b = wv % nbe, sub = wv / nbe, r0 = sub * P2_R;           // this wave: batch b, rows r0 .. r0 + P2_R - 1 of a chunk
    const bool have = b < nbatch;
    const ColFacts f = col_facts(a, 64 * b + lane);            // the wave's batch is the same in every chunk
    // per chunk, per lane: ro = row_off[first row of the wave + lane] for lane <= rows of the wave; lanes 62 / 63: the chunk's
    // first and end offsets.  pre = batch_pre[row][b] for lane < rows of the wave.
    auto chunk_rows = [&](uint32_t crow0) { return a.nrows - crow0 < CR ? a.nrows - crow0 : CR; };
    auto wave_rows = [&](uint32_t crows) { return crows > r0 ? (crows - r0 < P2_R ? crows - r0 : P2_R) : 0u; };
    auto load_ro = [&](uint32_t crow0, uint32_t crows, uint32_t rows) -> uint64_t {
        if (lane == 62) return a.row_off[crow0];
        if (lane == 63) return a.row_off[crow0 + crows];
        return (rows && lane <= rows) ? a.row_off[crow0 + r0 + lane] : 0ull;
    };
    uint32_t at0;
    uint64_t ro;
    uint32_t pre;
    {
        const uint32_t crow0 = blockIdx.x * CR, crows = chunk_rows(crow0), rows = wave_rows(crows);
        at0 = stage_chunk(a, stage, crow0, crows);
        ro = load_ro(crow0, crows, rows);
        pre = (have && lane < rows) ? a.batch_pre[(uint64_t)(crow0 + r0 + lane) * nbatch + b] : 0u;
        __syncthreads();
    }
    for (uint32_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
        const uint32_t crow0 = chunk * CR, crows = chunk_rows(crow0);
        const uint32_t row0 = crow0 + r0, rows = wave_rows(crows);
        const uint64_t base = ((uint64_t)lane_value((uint32_t)(ro >> 32), 62) << 32) | lane_value((uint32_t)ro, 62);
        const uint32_t nbytes = lane_value((uint32_t)ro, 63) - (uint32_t)base;        // a chunk is far below 4 GiB
        const uint32_t shift = (uint32_t)base & 15u;
        const uint32_t my_at = (uint32_t)ro - (uint32_t)base + shift, my_pre = pre;
        uint64_t v[P2_R + 1];
        load_column_staged<P2_R>(stage, at0 + r0 * a.rec_stride, a.rec_stride, f, row0, rows, v);
        // request the next chunk's records, row offsets and prefixes
        const uint32_t next = chunk + gridDim.x;
        const bool more = next < nchunks;
        const uint32_t ncrow0 = more ? next * CR : crow0;
        const uint32_t ncrows = chunk_rows(ncrow0), nrows_w = wave_rows(ncrows);
        const uint32_t nfirst = ncrow0 ? ncrow0 - 1 : 0;
        const uint8_t *src = a.rec + (uint64_t)nfirst * a.rec_stride;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15u);
        const uint32_t sbytes = more ? mis + (ncrow0 + ncrows - nfirst) * a.rec_stride : 0u;
        const uint8_t *src16 = src - mis;
        uint4 g0 = make_uint4(0, 0, 0, 0), g1 = g0, g2 = g0, g3 = g0;
        const uint32_t o0 = threadIdx.x * 16u;
        if (o0 < sbytes) g0 = stream_load16(src16 + o0);
        if (o0 + 4096u < sbytes) g1 = stream_load16(src16 + o0 + 4096u);
        if (o0 + 8192u < sbytes) g2 = stream_load16(src16 + o0 + 8192u);
        if (o0 + 12288u < sbytes) g3 = stream_load16(src16 + o0 + 12288u);
        ro = more ? load_ro(ncrow0, ncrows, nrows_w) : 0ull;
        pre = (more && have && lane < nrows_w) ? a.batch_pre[(uint64_t)(ncrow0 + r0 + lane) * nbatch + b] : 0u;

        if (have && rows) emit_batch<P2_R>(a, p2_lds, b, lane, f, v, row0, rows, my_at, my_pre, nflag);
        lds_barrier();          // the image is complete; every wave has read its columns from the staging buffer
        if (o0 < sbytes) *reinterpret_cast<uint4 *>(stage + o0) = g0;
        if (o0 + 4096u < sbytes) *reinterpret_cast<uint4 *>(stage + o0 + 4096u) = g1;
        if (o0 + 8192u < sbytes) *reinterpret_cast<uint4 *>(stage + o0 + 8192u) = g2;
        if (o0 + 12288u < sbytes) *reinterpret_cast<uint4 *>(stage + o0 + 12288u) = g3;
        at0 = mis + (ncrow0 - nfirst) * a.rec_stride;
        copy_image_out(a, p2_lds, base, shift, nbytes);
        lds_barrier();          // the image has been read, the next chunk's records are in place
    }
}

// ------------------------------------------------------------------------------------------
// rows of mostly empty fields: a wave per row, going by the map of non-zero records (Pass2Args.nz)
// ------------------------------------------------------------------------------------------
// Pass 1 left rec valid only where the map (a bit per field) is set.  A column can change between row r - 1 and row r only where one of
// the two has a record: those columns (in column order) are gathered from the two rows of the map into LDS, and only they are
// looked at -- tens of the thousands of columns such tables have.  Flags, stored values and their order are what the column
// kernels above produce: flag = the raw values differ (first row: first_row_flag), value = stored_from_raw, in column order.
constexpr uint32_t P2S_WAVES = 4;
constexpr uint32_t P2S_ROWS = 8;
constexpr uint32_t P2S_IMG = 1024;      // emit: a row's output of up to about this many bytes is put together in LDS and leaves in 16-byte stores
__host__ __device__ inline uint32_t p2s_cand_bytes(uint32_t nused) { return (nused * 2u + 15u) & ~15u; }
__host__ __device__ inline uint32_t p2s_img_bytes(uint32_t nused, bool emit) { return emit ? ((((nused + 7u) >> 3) + 15u) & ~15u) + 16u + P2S_IMG : 0u; }

constexpr uint32_t P2S_PIECES = 2;      // PRE: map rows of up to 4096 columns are read one row ahead (a word of 32 columns per lane and piece)

template <bool EMIT, bool PRE>
__global__ __launch_bounds__(64 * P2S_WAVES) void k_pass2_rows_sparse(const Pass2Args a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t p2s_lds[];
    const uint32_t lane = lane_id(), wv = threadIdx.x >> 6;
    const uint32_t nflag = (a.nused + 7) >> 3;
    // per wave: the candidates (u16 each, at most nused: column | record in this row << 14 | in the row above << 15), then (emit)
    // the row's image: flag bytes, values
    const uint32_t imgcap = p2s_img_bytes(a.nused, EMIT);
    const uint32_t per_wave = p2s_cand_bytes(a.nused) + imgcap;
    uint16_t *cand = reinterpret_cast<uint16_t *>(p2s_lds + (size_t)wv * per_wave);
    uint8_t *fimg = p2s_lds + (size_t)wv * per_wave + p2s_cand_bytes(a.nused);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    // PRE: the map row is read while the row before it is worked on; its word per piece stays in a register and is "the row
    // above" of the next row
    auto load_row = [&](uint64_t r, uint32_t (&v)[P2S_PIECES]) {
        const uint32_t *m = a.nz + r * a.nz_stride;
#pragma unroll
        for (uint32_t k = 0; k < P2S_PIECES; ++k) {
            const uint32_t w = k * 64u + lane;
            v[k] = w < a.nz_stride ? m[w] : 0u;
        }
    };
    // (a wave takes P2S_ROWS rows in a row: the map row above is the one it has just read, its output is one run of bytes)
    for (uint64_t g0 = wave * P2S_ROWS; g0 < a.nrows; g0 += nwaves * P2S_ROWS) {
    uint32_t bcur[P2S_PIECES] = {}, bprev[P2S_PIECES] = {}, xn[P2S_PIECES] = {};
    if (PRE) {
        load_row(g0, bcur);
        if (g0) load_row(g0 - 1, bprev);
    }
    for (uint32_t r = 0; r < P2S_ROWS && g0 + r < a.nrows; ++r) {
        const uint64_t row = g0 + r;
        const bool more = r + 1 < P2S_ROWS && row + 1 < a.nrows;
        if (PRE && more) load_row(row + 1, xn);
        const uint32_t *m1 = a.nz + row * a.nz_stride, *m0 = row ? m1 - a.nz_stride : nullptr;
        const bool all = row == 0 && !a.prev_is_zero;      // a shard's first row is compared with the halo row: every used column
        uint32_t cnt = 0;
        if (all) {
            for (uint32_t u = lane; u < a.nused; u += 64) { const uint32_t c = a.used_col[u]; cand[u] = (uint16_t)(c | (((m1[c >> 5] >> (c & 31u)) & 1u) << 14)); }
            cnt = a.nused;
        } else {
            // 32 columns of the two map rows per lane and piece
            auto compact = [&](uint32_t c0, uint32_t b1, uint32_t b0) {
                uint32_t bits = b1 | b0;
                uint32_t total;
                uint32_t at = cnt + wave_excl_sum((uint32_t)__popc(bits), lane, total);
                while (bits) {
                    const uint32_t b = (uint32_t)__ffs((int)bits) - 1u;
                    bits &= bits - 1u;
                    // (the column, and which of the two rows has a record there: bits 14 and 15)
                    if (at < a.nused) cand[at] = (uint16_t)((c0 + b) | ((b1 >> b & 1u) << 14) | ((b0 >> b & 1u) << 15));
                    ++at;
                }
                cnt += total;
            };
            if (PRE) {
#pragma unroll
                for (uint32_t k = 0; k < P2S_PIECES; ++k)
                    if (k * 64u < a.nz_stride) compact((k * 64u + lane) * 32u, bcur[k], bprev[k]);
            } else {
                for (uint32_t wb = 0; wb < a.nz_stride; wb += 64u) {
                    const uint32_t w = wb + lane;
                    uint32_t x = 0, y = 0;
                    if (w < a.nz_stride) {
                        x = m1[w];
                        if (m0) y = m0[w];
                    }
                    compact(w * 32u, x, y);
                }
            }
            cnt = cnt < a.nused ? cnt : a.nused;
        }
        const uint64_t out0 = EMIT ? a.row_off[row] : 0;
        // emit: the image starts `shift` bytes into its buffer, so that 16-byte pieces of it are 16-byte pieces of the output;
        // a row too long for the buffer has only its flag bytes there and its values written where they belong
        uint32_t shift = 0;
        bool img = false;
        if (EMIT) {
            const uint32_t L = (uint32_t)(a.row_off[row + 1] - out0);
            shift = (uint32_t)((reinterpret_cast<uintptr_t>(a.out) + out0) & 15u);
            img = shift + L <= imgcap;
            if (!img) shift = 0;
            for (uint32_t k = lane * 4u; k < shift + nflag; k += 256u) *reinterpret_cast<uint32_t *>(fimg + k) = 0u;
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t run = nflag;
        // 128 candidates at a time; per candidate a chain of three loads: its column's facts and the two map bytes, the records,
        // the dictionary offset of a string
        for (uint32_t g = 0; g < cnt; g += 128) {
            bool ok[2], text[2], flag[2];
            uint32_t col[2], u[2], sz[2], n1[2], n0[2];
            uint64_t meta[2], base[2], cur[2], prev[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                ok[j] = g + 64u * j + lane < cnt;
                const uint32_t e = ok[j] ? cand[g + 64u * j + lane] : 0u;
                col[j] = e & 0x3fffu; n1[j] = e >> 14 & 1u; n0[j] = e >> 15;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                meta[j] = ok[j] ? a.col_meta[col[j]] : 0xFFFFFFull;      // which used column (24 bits), kind (2), stored size (4), record offset (32)
                base[j] = ok[j] ? a.col_meta[a.nz_cols + col[j]] : 0ull;   // the column's base value
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                u[j] = (uint32_t)meta[j] & 0xFFFFFFu;
                ok[j] = ok[j] && u[j] != 0xFFFFFFu;
                text[j] = ((uint32_t)meta[j] >> 24 & 3u) == K_TEXT;
                const uint32_t off = (uint32_t)(meta[j] >> 32);
                // a record is read only where it decides something: the map says it is not zero, so beside an empty field of the
                // other row it is a change whatever it holds (the emit kernel wants the value itself)
                cur[j] = (ok[j] && n1[j] && (EMIT || n0[j] || row == 0)) ? raw_value(a, (uint32_t)row, off, text[j]) : 0ull;
                prev[j] = (ok[j] && n1[j] && n0[j]) ? raw_value(a, (uint32_t)row - 1u, off, text[j]) : 0ull;
            }
            uint64_t x[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                flag[j] = ok[j] && ((n1[j] != 0u) != (n0[j] != 0u) || cur[j] != prev[j]);
                if (row == 0) flag[j] = ok[j] && first_row_flag(a, cur[j], u[j], text[j]);
                sz[j] = flag[j] ? ((uint32_t)meta[j] >> 26 & 15u) : 0u;
                x[j] = 0ull;
                if (EMIT && flag[j] && cur[j]) x[j] = text[j] ? (uint64_t)a.id_off[(uint32_t)cur[j] - 1u] : cur[j] - base[j];
            }
            uint32_t t0, t1;
            uint32_t at[2];
            at[0] = wave_excl_sum(sz[0], lane, t0);
            at[1] = t0 + wave_excl_sum(sz[1], lane, t1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (EMIT && flag[j]) {
                    const uint32_t fb = shift + (u[j] >> 3);      // bit u of the flag bytes, through the aligned word that holds it
                    atomicOr(reinterpret_cast<uint32_t *>(fimg + (fb & ~3u)), 1u << ((fb & 3u) * 8u + (u[j] & 7u)));
                    uint64_t v = x[j];
                    const uint32_t w = sz[j];
                    if (img) {
                        uint8_t *q = fimg + shift + run + at[j];
                        if (w == 8) { reinterpret_cast<PackedW64 *>(q)->v = v; }
                        else {
                            if (w & 4u) { reinterpret_cast<PackedW32 *>(q)->v = (uint32_t)v; q += 4; v >>= 32; }
                            if (w & 2u) { reinterpret_cast<PackedW16 *>(q)->v = (uint16_t)v; q += 2; v >>= 16; }
                            if (w & 1u) *q = (uint8_t)v;
                        }
                    } else {
                        uint8_t *q = a.out + out0 + run + at[j];
                        if (w == 8) { reinterpret_cast<PackedW64 *>(q)->v = v; }
                        else {
                            if (w & 4u) { reinterpret_cast<PackedW32 *>(q)->v = (uint32_t)v; q += 4; v >>= 32; }
                            if (w & 2u) { reinterpret_cast<PackedW16 *>(q)->v = (uint16_t)v; q += 2; v >>= 16; }
                            if (w & 1u) *q = (uint8_t)v;
                        }
                    }
                }
            }
            run += t0 + t1;
        }
        if (EMIT) {
            __builtin_amdgcn_wave_barrier();
            if (img) {
                uint8_t *gdst = a.out + (out0 - shift);              // 16-byte aligned
                const uint32_t endb = shift + run;
                for (uint32_t o = lane * 16u; o < endb; o += 1024u) {
                    if (o >= shift && o + 16u <= endb) {
                        *reinterpret_cast<uint4 *>(gdst + o) = *reinterpret_cast<const uint4 *>(fimg + o);
                    } else {
                        const uint32_t lo = o > shift ? o : shift, hi = o + 16u < endb ? o + 16u : endb;
                        for (uint32_t k = lo; k < hi; ++k) gdst[k] = fimg[k];
                    }
                }
            } else {
                for (uint32_t k = lane; k < nflag; k += 64) a.out[out0 + k] = fimg[k];
            }
            __builtin_amdgcn_wave_barrier();
        } else if (lane == 0) {
            a.row_len[row] = run;
        }
        if (PRE) {
#pragma unroll
            for (uint32_t k = 0; k < P2S_PIECES; ++k) { bprev[k] = bcur[k]; if (more) bcur[k] = xn[k]; }
        }
    }
    }
}

template <bool EMIT>
static void launch_pass2_sparse(const Pass2Args &a, hipStream_t s)
{
    const size_t per_wave = (size_t)p2s_cand_bytes(a.nused) + p2s_img_bytes(a.nused, EMIT);
    const size_t lds = per_wave * P2S_WAVES;
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const uint64_t want = ((uint64_t)a.nrows + P2S_WAVES * P2S_ROWS - 1) / (P2S_WAVES * P2S_ROWS);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)cus * 8);
    if (a.nz_stride <= 64u * P2S_PIECES && !experiment("ZDW_P2S_NO_PRE")) hipLaunchKernelGGL((k_pass2_rows_sparse<EMIT, true>), dim3(grid), dim3(64 * P2S_WAVES), lds, s, a);
    else hipLaunchKernelGGL((k_pass2_rows_sparse<EMIT, false>), dim3(grid), dim3(64 * P2S_WAVES), lds, s, a);
}

template <bool EMIT>
static void launch_pass2(const Pass2Args &a, hipStream_t s)
{
    if (!a.nrows) return;
    if (a.nz) { launch_pass2_sparse<EMIT>(a, s); return; }
    uint32_t g = 1;
    while (g < 64 && g < a.nused) g <<= 1;
    // Which kernels: the lane = column kernels need not be full to win.  Measured per 4 GB, G-lanes-per-row kernels vs column
    // kernels: 32 columns 1.9 + 3.0 vs 0.85 + 2.1 ms (size + emit), 20: 2.9 + 4.5 vs 1.1 + 3.1, 16: 1.9 + 3.0 vs 1.4 + 3.8,
    // 9: 3.2 + 4.8 vs 2.2 + 6.3, 6: 2.7 + 3.8 vs 3.4 + 9.3.
    uint32_t min_cols = EMIT ? 17u : 8u;
    if (const char *e = test_hook("ZDW_P2_COLS_MIN")) min_cols = (uint32_t)atoi(e);      // (A/B measurements)
    // (the emit kernel's LDS image: 8 rows of output, or 4 when rows are very wide; the size kernel has none)
    const uint32_t size_tot_lds = P2_R_SIZE * p2_waves_per_batch((a.nused + 63) / 64) * ((a.nused + 63) / 64) * 2u;      // size kernel: u16 per (row, batch)
    if (a.nused >= min_cols && a.batch_pre && (EMIT ? 4u * a.out_row_max + 32u <= P2_LDS_BUDGET : size_tot_lds <= P2_LDS_BUDGET)) {
        // wide schema: persistent grids sized to what is resident
        int dev = 0, cus = 256, occ = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const uint32_t nbatch = (a.nused + 63) / 64;
        const bool direct = test_hook("ZDW_P2_DIRECT") != nullptr;      // (the variables here: A/B measurements)
        if (!EMIT) {
            const uint32_t CR = P2_R_SIZE * p2_waves_per_batch(nbatch);          // rows per chunk
            const bool staged = !direct && p2_stage_bytes(CR, a.rec_stride) <= P2_STAGE_BUDGET;
            const uint64_t want = ((uint64_t)a.nrows + CR - 1) / CR;
            const uint32_t lds = ((CR * nbatch * 2u + 15u) & ~15u) + (staged ? p2_stage_bytes(CR, a.rec_stride) : 0u);
            const void *fn = staged ? (const void *)k_pass2_size_cols<true> : (const void *)k_pass2_size_cols<false>;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, fn, 256, lds) != hipSuccess || occ < 1) occ = 1;
            uint32_t per_cu = std::min(6u, (uint32_t)occ);      // six workgroups per CU measured best (8: 1.36x the time)
            // ... alone on the chip.  Beside the dictionary sort (second stream, ensure_dict) fewer leave the sort's launch-bound
            // kernels room: 10 GB, sort / sizes / step 3.5 / 1.6 / 20.0 ms with six (no better than one after the other),
            // 2.9 / 2.1 / 19.35 with five, 2.7 / 2.5 / 19.08 with four, 2.7 / 3.3 / 19.14 with three, 2.4 / 4.2 / 19.9 with two
            // (tools/ab_size_wgs.sh; three was best while the sort took 3.1 ms)
            if (a.background) per_cu = std::min(per_cu, 4u);
            if (const char *e = experiment("ZDW_P2_SIZE_WGS")) per_cu = std::max(1u, (uint32_t)atoi(e));
            const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)cus * per_cu);
            if (staged) hipLaunchKernelGGL(k_pass2_size_cols<true>, dim3(grid), dim3(256), lds, s, a);
            else hipLaunchKernelGGL(k_pass2_size_cols<false>, dim3(grid), dim3(256), lds, s, a);
        } else {
            uint32_t R = 8;
            if (8u * a.out_row_max + 32u > P2_LDS_BUDGET) R = 4;     // very wide rows: fewer of them per chunk
            // (1000 columns, 4 GB: 8 rows straight from HBM 2.08 ms, 4 rows staged 2.78, G-lanes-per-row kernel 2.52)
            if (const char *e = test_hook("ZDW_P2_EMIT_ROWS")) R = atoi(e) == 16 ? 16u : (atoi(e) == 4 ? 4u : 8u);
            uint32_t mode = 0, CR = R;                                  // rows per chunk
            if (!direct && !experiment("ZDW_P2_EMIT_DIRECT") && p2_stage_bytes(R, a.rec_stride) <= P2_STAGE_BUDGET) {
                mode = 1;
                const uint32_t CRp = R * p2_waves_per_batch(nbatch);
                if (nbatch <= 4 && p2_stage_bytes(CRp, a.rec_stride) <= 4u * 4096u && CRp * a.out_row_max + 32u <= P2_LDS_BUDGET &&
                    !test_hook("ZDW_P2_EMIT_NOPIPE")) { mode = 2; CR = CRp; }
            }
            const uint64_t want = ((uint64_t)a.nrows + CR - 1) / CR;
            const uint32_t lds = ((CR * a.out_row_max + 32u + 15u) & ~15u) + (mode ? p2_stage_bytes(CR, a.rec_stride) : 0u);
            typedef void (*EmitFn)(const Pass2Args);
            static const EmitFn fns[3][3] = {{k_pass2_emit_cols<false, 8>, k_pass2_emit_cols<true, 8>, k_pass2_emit_cols_piped<8>},
                                             {k_pass2_emit_cols<false, 16>, k_pass2_emit_cols<true, 16>, k_pass2_emit_cols_piped<16>},
                                             {k_pass2_emit_cols<false, 4>, k_pass2_emit_cols<true, 4>, k_pass2_emit_cols_piped<4>}};
            const EmitFn fn = fns[R == 16 ? 1 : (R == 4 ? 2 : 0)][mode];
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)fn, 256, lds) != hipSuccess || occ < 1) occ = 1;
            // measured (10 GB x 200 columns): staged 8-row chunks 5 / 6 / 7 / 8 workgroups per CU = 5.5 / 4.9 / 4.5 / 6.7 ms
            // (rows read straight from HBM -- 1000 columns, 8 GB with ids in every 20th: 2 / 3 / 4 / 5 / 6 / 7 / 8 / 10 / 12 / 16 workgroups
            // per CU = 8.0 / 5.7 / 4.5 / 6.7 / 5.8 / 5.1 / 4.6 / 5.4 / 4.6 / 4.6 ms, five of them resident: multiples of four, whatever
            // is resident -- tools/ab_c5_pass2.sh; it was min(6, resident) = 5)
            uint32_t per_cu = mode ? std::min(7u, (uint32_t)occ) : ((uint32_t)occ >= 4u ? 4u : (uint32_t)occ);
            if (const char *e = experiment("ZDW_P2_WGS")) per_cu = std::max(1u, (uint32_t)atoi(e));
            const uint32_t grid = (uint32_t)std::min<uint64_t>(want, (uint64_t)cus * per_cu);
            hipLaunchKernelGGL(fn, dim3(grid), dim3(256), lds, s, a);
        }
        return;
    }
    const uint64_t rows_per_block = (256 / 64) * (64 / g);
    bool loop = a.nrows >= (1u << 16);                     // (small blocks: the plain kernel's more waves win)
    if (const char *e = test_hook("ZDW_P2_LOOP")) loop = atoi(e) != 0;      // tests
    if (loop && a.nused <= g) {
        const uint64_t per_block = rows_per_block * P2N_ITERS;
        const uint32_t grid = (uint32_t)((a.nrows + per_block - 1) / per_block);
        const uint32_t lds = 4u * p2n_image_bytes(g, a.out_row_max);
        bool stage = EMIT && lds <= 48u * 1024u;
        if (const char *e = test_hook("ZDW_P2_LOOP_STAGE")) stage = EMIT && atoi(e) != 0 && lds <= 60u * 1024u;      // tests
#define P2N_LAUNCH(GG)                                                                                                              \
        if (stage) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2_loop<GG, EMIT, EMIT>), dim3(grid), dim3(256), lds, s, a);              \
        else hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2_loop<GG, EMIT, false>), dim3(grid), dim3(256), 0, s, a)
        switch (g) {
        case 1: P2N_LAUNCH(1); break;
        case 2: P2N_LAUNCH(2); break;
        case 4: P2N_LAUNCH(4); break;
        case 8: P2N_LAUNCH(8); break;
        case 16: P2N_LAUNCH(16); break;
        case 32: P2N_LAUNCH(32); break;
        default: P2N_LAUNCH(64); break;
        }
#undef P2N_LAUNCH
        return;
    }
    const uint32_t grid = (uint32_t)((a.nrows + rows_per_block - 1) / rows_per_block);
    switch (g) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<1, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<2, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 4: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<4, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<8, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 16: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<16, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    case 32: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<32, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    default: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_pass2<64, EMIT>), dim3(grid), dim3(256), 0, s, a); break;
    }
}

// stored values of the block's last row (what the next shard of a row-sharded block compares its
// first row with); independent of the flags, so it needs no pass over the rows
__global__ __launch_bounds__(256) void k_last_row(const Pass2Args a)
{
    const uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < a.nused) a.last_stored[u] = stored_value(a, a.nrows - 1, u);
}

void launch_last_row(const Pass2Args &a, hipStream_t s)
{
    if (!a.nrows || !a.nused) return;
    hipLaunchKernelGGL(k_last_row, dim3((a.nused + 255) / 256), dim3(256), 0, s, a);
}

void launch_pass2_size(const Pass2Args &a, hipStream_t s) { launch_pass2<false>(a, s); }
void launch_pass2_emit(const Pass2Args &a, hipStream_t s) { launch_pass2<true>(a, s); }

}  // namespace zdw
