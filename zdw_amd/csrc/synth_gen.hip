// synth_gen.hip — device generator of the benchmark's synthetic TSV: writes rows
// [row0, row0+nrows) of the table straight into HBM.  Benchmark/test data plumbing, not part of the
// encode path.  Field (row, col) is a pure function of (seed, row, col): a field repeats the row
// above with p = 0.3 (walk up to the first non-repeating row), is empty with p = 0.25, else is
// drawn by column class:
//   0 word   : Zipf(1.1) rank over `vocab` words, word i = 4..24 chars of [a-z0-9_] from hash(i)
//   1..4 int : uniform in [0, 2^7 | 2^15 | 2^31 | 2^47)
//   5 char(1), 6 datetime "2024-01-DD HH:MM:SS", 7 decimal "<int>.<12 digits>"
//   8 id     : 24-character token from the cell's own hash (C5's high-cardinality varchar(64) columns)
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
    x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
    x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t cell_hash(uint64_t seed, uint64_t row, uint32_t col, uint32_t salt)
{
    return mix(seed ^ mix(row * 0x9E3779B97F4A7C15ull + col) ^ ((uint64_t)salt << 56));
}

template <bool WRITE>
__device__ inline uint32_t put_u64(uint8_t *q, uint64_t v)
{
    char tmp[20];
    uint32_t n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    if (WRITE) for (uint32_t i = 0; i < n; ++i) q[i] = (uint8_t)tmp[n - 1 - i];
    return n;
}
template <bool WRITE>
__device__ inline uint32_t put_fixed(uint8_t *q, uint64_t v, uint32_t digits)
{
    if (WRITE) for (uint32_t i = 0; i < digits; ++i) { q[digits - 1 - i] = (uint8_t)('0' + v % 10); v /= 10; }
    return digits;
}

// `vocab_and_len`: bits 0..23 = vocabulary size; bits 24..31 = how far the longest word exceeds 24 characters (0 = the
// benchmark's table: 4..24).  g_empty16: a field is empty when the low 16 bits of its hash are below it (16384 = the
// benchmark's 25 %; set per launch for tables of other shapes).
__device__ uint32_t g_empty16 = 16384;
__device__ uint32_t g_dead16 = 0;          // a column is empty in every row when the low 16 bits of a hash of its index are below this

template <bool WRITE>
__device__ inline uint32_t emit_field(uint8_t *q, uint64_t seed, uint64_t row, uint32_t col, uint32_t gen, uint32_t vocab_and_len)
{
    const uint32_t vocab = vocab_and_len & 0xFFFFFFu, span = 21u + (vocab_and_len >> 24);
    // repeat chain: walk up while this field repeats the row above
    uint32_t steps = 0;
    while (row > 0 && steps < 64 && (cell_hash(seed, row, col, 1) & 0xFFFF) < 19661) { --row; ++steps; }   // 0.3 * 65536
    const uint64_t h = cell_hash(seed, row, col, 2);
    if (g_dead16 && (mix(seed ^ (0xD1B54A32D192ED03ull * (col + 1))) & 0xFFFF) < g_dead16) return 0;          // a column nobody fills
    if ((h & 0xFFFF) < g_empty16) return 0;                                                                  // 25 % empty
    const uint64_t r = h >> 16;
    switch (gen) {
    case 0: {
        const float u = (float)(r & 0xFFFFFF) * (1.0f / 16777216.0f);
        const float s1 = -0.1f;                                   // 1 - s, s = 1.1
        const float top = __powf((float)vocab + 1.0f, s1) - 1.0f;
        float x = __powf(1.0f + u * top, 1.0f / s1);
        uint32_t rank = (uint32_t)x;
        if (rank < 1) rank = 1;
        if (rank > vocab) rank = vocab;
        uint64_t w = mix(seed ^ (0x9E3779B97F4A7C15ull * (uint64_t)rank));
        const uint32_t len = 4 + (uint32_t)(w % span);
        for (uint32_t j = 0; j < len; ++j) {
            w = mix(w + j + 1);
            if (WRITE) q[j] = (uint8_t)"abcdefghijklmnopqrstuvwxyz0123456789_"[w % 37];
        }
        return len;
    }
    case 8: {   // id-like token: 24 characters drawn from this cell's own hash (practically unique: the high-cardinality columns of C5)
        uint64_t w = mix(h ^ 0xA24BAED4963EE407ull);
        for (uint32_t j = 0; j < 24; ++j) {
            if (j == 12) w = mix(w + h);
            if (WRITE) q[j] = (uint8_t)"abcdefghijklmnopqrstuvwxyz0123456789_"[(w >> (5 * (j % 12))) % 37];
        }
        return 24;
    }
    case 1: return put_u64<WRITE>(q, r & 0x7Full);
    case 2: return put_u64<WRITE>(q, r & 0x7FFFull);
    case 3: return put_u64<WRITE>(q, r & 0x7FFFFFFFull);
    case 4: return put_u64<WRITE>(q, r & 0x7FFFFFFFFFFFull);
    case 5: if (WRITE) q[0] = (uint8_t)('a' + r % 26); return 1;
    case 6: {
        const uint32_t t = (uint32_t)(r % (86400u * 28u));
        if (WRITE) {
            q[0] = '2'; q[1] = '0'; q[2] = '2'; q[3] = '4'; q[4] = '-'; q[5] = '0'; q[6] = '1'; q[7] = '-';
            put_fixed<true>(q + 8, 1 + t / 86400, 2); q[10] = ' ';
            put_fixed<true>(q + 11, t / 3600 % 24, 2); q[13] = ':';
            put_fixed<true>(q + 14, t / 60 % 60, 2); q[16] = ':';
            put_fixed<true>(q + 17, t % 60, 2);
        }
        return 19;
    }
    default: {
        const uint32_t n = put_u64<WRITE>(q, r % 100000);
        if (WRITE) q[n] = '.';
        put_fixed<WRITE>(q + n + 1, mix(r) % 1000000000000ull, 12);
        return n + 13;
    }
    }
}

__global__ __launch_bounds__(256) void k_synth_lengths(const uint8_t *__restrict__ gens, uint32_t ncols, uint64_t seed, uint64_t row0,
                                                        uint64_t nrows, uint32_t vocab, int64_t *__restrict__ lens)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    uint64_t n = ncols;   // tabs + newline
    for (uint32_t c = 0; c < ncols; ++c) n += emit_field<false>(nullptr, seed, row0 + i, c, gens[c], vocab);
    lens[i] = (int64_t)n;
}

__global__ __launch_bounds__(256) void k_synth_fill(const uint8_t *__restrict__ gens, uint32_t ncols, uint64_t seed, uint64_t row0,
                                                     uint64_t nrows, uint32_t vocab, const int64_t *__restrict__ offs,
                                                     uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows) return;
    uint8_t *q = out + offs[i];
    for (uint32_t c = 0; c < ncols; ++c) {
        q += emit_field<true>(q, seed, row0 + i, c, gens[c], vocab);
        *q++ = c + 1 == ncols ? '\n' : '\t';
    }
}

}  // namespace

// fraction of empty fields for the launches that follow, in 1/65536 (16384 = the benchmark's table)
extern "C" int zdw_synth_set_empty(uint32_t empty16, uint32_t dead16)
{
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_empty16), &empty16, sizeof empty16);
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(g_dead16), &dead16, sizeof dead16);
    return (int)e;
}

extern "C" int zdw_synth_row_lengths(const void *gens, uint32_t ncols, uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t vocab,
                                     void *lens, void *stream)
{
    if (!nrows) return 0;
    hipLaunchKernelGGL(k_synth_lengths, dim3((uint32_t)((nrows + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t *)gens, ncols, seed, row0, nrows, vocab, (int64_t *)lens);
    return (int)hipGetLastError();
}

extern "C" int zdw_synth_fill(const void *gens, uint32_t ncols, uint64_t seed, uint64_t row0, uint64_t nrows, uint32_t vocab,
                              const void *offs, void *out, void *stream)
{
    if (!nrows) return 0;
    hipLaunchKernelGGL(k_synth_fill, dim3((uint32_t)((nrows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)gens,
                       ncols, seed, row0, nrows, vocab, (const int64_t *)offs, (uint8_t *)out);
    return (int)hipGetLastError();
}
