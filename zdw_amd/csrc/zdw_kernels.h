// zdw_kernels.h — launch-level interface between the C ABI (zdw_capi.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace zdw {

// position space of dictionary strings: [0, 2^39) = the TSV image, [2^39, 2^40) = imported strings
constexpr uint64_t POS_BITS = 40;
constexpr uint64_t POS_MASK = (1ull << POS_BITS) - 1ull;
constexpr uint64_t EXTRA_BASE = 1ull << 39;
constexpr uint32_t COL_SKIP = 0xFFFFFFu;

// Written by the kernels, read by the host after each launch.  The two busy counters sit on memory lines of
// their own: every wave polls `overflow`, and an atomic that shares a line with a polled word queues behind
// the polls (measured: ~40 us per add instead of ~2).
struct P1Status {
    uint32_t bad_row;       // smallest 0-based row with a wrong field count, 0xFFFFFFFF = none
    uint32_t overflow;      // dictionary table (or the list of uniques) too small: retry with more slots
    uint32_t long_count;    // rows >= 16383 bytes recorded in long_rows
    uint32_t nul_seen;      // 1: NUL byte in the input (unsupported); 2: a single field of 1 GiB or more
    uint64_t split_pos;     // start of row `row_limit` (the first row of the next block)
    uint32_t max_len;       // longest string beyond 24 bytes (bounds the sort rounds)
    uint32_t pad;           // profiling only
    uint32_t deferred_lost; // the deferred-field queue was too small: redo the pass with the all-in-one kernel
    uint32_t pad2;
    uint8_t gap0[256 - 40];
    uint32_t entries;       // unique strings inserted
    uint8_t gap1[256 - 4];
    uint32_t id_next;       // unique ids handed out so far (ids of lost races and unused chunk tails stay holes: length 0)
    uint8_t gap2[256 - 4];
    uint32_t deferred;      // fields queued for the general-form kernel
    uint8_t gap3[256 - 4];
    uint32_t next_span;     // pass 1 counting its own rows (Pass1Args.span_state): the next span to hand out
    uint32_t fold_abort;    // ... a wave gave up waiting for the spans before its own (cannot happen; never hang on it)
    uint64_t rows_total;    // ... rows that start in the block, + 1 (0 = the last span was not reached)
    uint8_t gap4[256 - 16];
    uint32_t next_group;    // pass 1 with a lane per row (k_rows.hip): the next group of rows to hand out
    uint8_t gap5[256 - 4];
    unsigned long long dbg[16];     // profiling builds only (-DZDW_EXPERIMENTS): cycles per part of a kernel, summed over its waves
    uint8_t gap6[256 - 128];
};

struct LongRow { uint64_t start; uint64_t len; };   // len includes the terminating newline

struct Pass1Args {
    const uint8_t *in; uint64_t n;        // TSV image and its allocation size
    uint64_t lo, hi;                      // block start; effective end (last row terminator + 1)
    uint64_t span0; uint32_t nspans;      // spans [span0, span0 + nspans) cover [lo, hi)
    uint32_t span_shift;                  // a span = 2^span_shift bytes
    const uint32_t *span_rowbase;         // exclusive scan of per-span row starts
    uint64_t *span_state;                 // not null: no row scan has run -- the waves take the spans in order (a ticket each), count the
                                          // rows that start in theirs and learn their first row's number from the spans before (a word per
                                          // span: 1 << 62 | count, then 2 << 62 | count of all spans up to it; zeroed before the launch)
    const uint8_t *extra;                 // imported strings (multi-GPU merge), NUL terminated
    uint32_t ncols;
    const uint8_t *col_kind;
    const uint32_t *col_recoff;
    const int32_t *col_numidx;
    uint32_t nnum;
    uint32_t use_lds_ranges;              // 1: column facts, min/max and seen in workgroup LDS, flushed once per workgroup
    uint32_t rec_stride; uint8_t *rec;    // rec[row][...]: u64 per numeric column, u32 id per text column
    uint32_t row_limit, partial_fields;
    uint32_t trim;
    uint64_t *slots; uint64_t slot_mask; uint32_t max_entries;
    uint64_t *uniq_pos; uint32_t *uniq_len; uint32_t id_cap;   // per unique id: first occurrence and length (0 = hole)
    uint64_t *uniq_key;                   // optional: the string's first 24 bytes by id (three words each: uniq_key[3 * id + k]),
                                          // little-endian and zero beyond the string's end like the table slot's copy -- what the dictionary sort and the blob
                                          // read instead of going back to the image for every string
    uint64_t *col_min; uint64_t *col_max; uint8_t *col_seen;
    P1Status *st;
    LongRow *long_rows; uint32_t long_cap;
    uint32_t track_first;                 // uniq_pos[id] = the string's first occurrence (a minimum over all of them; the array starts as all ones): the memory-limit model
    uint32_t skip_empty;                  // tabs that directly follow a boundary (empty fields) get no descriptor (tables of mostly empty fields)
    uint32_t rebase_at;                   // relative position at which a visit moves its base (2^23: a descriptor holds 24 bits of position; tests lower it)
    uint64_t *dq; uint32_t dq_cap;        // deferred fields (two words each: position | column << 40, length | row << 32); dq == nullptr: none, one kernel does everything
    uint32_t rec_zeroed;                  // 1: rec was zeroed before the launch: empty fields leave nothing to do
    uint32_t *nz; uint32_t nz_stride;     // not null (with rec_zeroed, instead of a zeroed rec): a bit per field, bit (column & 31) of word nz[row * nz_stride + column / 32], set where a non-zero record was
                                          // written; what rec holds elsewhere means nothing (tables of mostly empty fields: pass 2 goes by this map)
    uint32_t dense_boundaries;            // 1: the input averages more boundaries per round than the small descriptor ring takes (tables of mostly empty fields): large ring
    uint32_t debug_skip;                  // read by profiling builds only (-DZDW_EXPERIMENTS, ZDW_DEBUG_SKIP): 1 no field work, 2 no dictionary, 4 no record stores, 32 the
                                          // tokeniser alone.  Always declared, and last: the struct's layout is the same whatever an object file was compiled with
};

struct Pass2Args {
    const uint8_t *rec; uint32_t rec_stride; uint32_t nrows;
    uint32_t nused;
    uint32_t out_row_max;          // most bytes one row can take: flag bytes + every used column's size
    const uint32_t *used_recoff;   // per used column: offset inside a rec row
    const uint8_t *used_kind;      // K_TEXT / K_INT / K_CHAR
    const uint8_t *used_size;      // bytes stored per value
    const uint64_t *used_base;     // column base (min - 1; 0 for text)
    const uint32_t *id_off;        // dictionary offset per unique id
    uint32_t nids;                 // entries of id_off
    const uint64_t *prev0;         // stored values of the row before row 0 (per used column)
    uint32_t prev_is_zero;         // 1: no row precedes row 0 (prev0 is all zero): its text flags are "the field is not empty", decided
                                   // without the dictionary (offsets are >= 1), so the row lengths can be computed while it is sorted
    uint32_t *row_len;             // out (size kernel)
    uint16_t *batch_pre;           // column kernels, size kernel -> emit kernel: per (row, 64-column batch) the value bytes of the batches before it
    const uint64_t *row_off;       // in (emit kernel): exclusive scan of row_len
    uint8_t *out;
    uint64_t *last_stored;         // k_last_row only: stored values of the last row (per used column)
    uint32_t background;           // size kernel: it runs beside the dictionary sort (second stream): a smaller grid
    const uint32_t *nz; uint32_t nz_stride;     // not null: rec is valid only where the map's bit is set (Pass1Args.nz; nz_stride words per row): the kernels for sparse rows
    const uint32_t *used_col;      // ... per used column: its column in the schema
    const uint64_t *col_meta;      // ... per column of the schema: used column (bits 0-23, 0xFFFFFF: none), kind (24-25), stored size (26-29), record offset (32-63)
    uint32_t nz_cols;              // ... columns of the schema; col_meta[nz_cols + column] = the column's base value
};

// pass 1 with a lane per row (k_rows.hip): the row index, then the walk
struct RowIndexArgs {
    const uint8_t *in; uint64_t n;        // TSV image and its allocation size
    uint64_t lo, hi;                      // block start; effective end (last row terminator + 1)
    uint64_t span0; uint32_t nspans;      // 16 KiB spans [span0, span0 + nspans) cover [lo, hi)
    uint64_t *span_state;                 // a word per span, zeroed (the look-back of the row count)
    uint64_t *rowidx;                     // out: [2 i] start of row i, [2 i + 1] position of its terminator; [2 rows] = hi
    uint64_t rows_cap;                    // rows the index has room for (rows_cap + 2 pairs allocated)
    uint8_t *bsmap;                       // out, per span: it holds a backslash
    P1Status *st;                         // next_span (zeroed), rows_total, nul_seen, fold_abort
};
struct RowsArgs {
    const uint64_t *rowidx;
    const uint8_t *bsmap; uint64_t bs_span0;      // bsmap[(position >> 14) - bs_span0]
    uint32_t rows_per_lane;                        // consecutive rows a lane walks before the wave takes the next group
    uint32_t find_only;                            // 1: the kernel looks strings up and inserts none (what it does not find at the first slot goes to the general form)
};
constexpr uint32_t RW_WPB_MAX = 6;                 // waves per workgroup (24 KiB of LDS each)

// ---- launchers (all asynchronous on `s`) ----
void launch_row_index(const RowIndexArgs &a, hipStream_t s);
void launch_pass1_rows(const Pass1Args &a, const RowsArgs &ra, hipStream_t s);
size_t pass1_rows_lds_bytes(uint32_t ncols, uint32_t nnum, uint32_t wpb);
void launch_find_effective_end(const uint8_t *in, uint64_t n, uint64_t lo, uint64_t *d_out, hipStream_t s);
void launch_count_row_starts(const uint8_t *in, uint64_t n, uint64_t lo, uint64_t hi, uint64_t span0, uint32_t nspans, uint32_t span_shift,
                             uint32_t *span_starts, P1Status *st, hipStream_t s);
void launch_pass1(const Pass1Args &a, hipStream_t s);
// rec[row][column] := 0 wherever the map's bit is clear (a record matrix that was written through Pass1Args.nz, for readers that want all of it)
void launch_rec_zero_fill(uint8_t *rec, uint32_t rec_stride, uint32_t nrows, uint32_t ncols, const uint8_t *col_kind, const uint32_t *col_recoff,
                          const uint32_t *nz, uint32_t nz_stride, hipStream_t s);
void launch_pass1_deferred(const Pass1Args &a, uint32_t count, hipStream_t s);
size_t pass1_lds_bytes(uint32_t ncols, uint32_t nnum, bool use_lds);

// exclusive scans; tmp must hold scan_tmp_bytes(n) bytes; total (optional) receives the sum
size_t scan_tmp_bytes(uint64_t n);
void launch_scan_u32_to_u32(const uint32_t *in, uint32_t *out, uint64_t n, void *tmp, uint64_t *total, hipStream_t s);
// two flags per element (bit 0, bit 1) -> their two exclusive scans (n + 1 elements each) in one go
void launch_scan_two_flags(const uint32_t *in, uint32_t *out0, uint32_t *out1, uint64_t n, void *tmp, hipStream_t s);
void launch_scan_u32_to_u64(const uint32_t *in, uint64_t *out, uint64_t n, void *tmp, uint64_t *total, hipStream_t s);

// dictionary
struct SortBuffers {   // all sized for n elements unless noted
    uint64_t *key_a, *key_b;
    uint32_t *val_a, *val_b;
    uint32_t *order;       // result: order[rank] = unique index
    uint32_t *group;       // tie-group id per position
    uint32_t *active_a, *active_b;   // positions still tied
    uint32_t *flags;       // scratch, n+1
    uint32_t *hist;        // 256 * sort_blocks(n)
    void *scan_tmp;
    uint32_t *counter;     // small device scratch (>= 4 u32)
};
uint32_t sort_blocks(uint64_t n);
constexpr uint32_t SORT_SPARE_WORDS = 8 + 4096 * 8 + 256 * 18;      // behind the histograms: dict_sort's flags and counts, its sample (4096 x 32 B), the packed key's table (256 x 72 B)
// sorts unique strings into strcmp order; returns through `order`.  Host-driven loop (synchronises on s).
int dict_sort(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len, uint64_t n,
              SortBuffers &b, hipStream_t s, uint32_t *rounds_out, uint32_t max_len, const uint64_t *uniq_key = nullptr);
void launch_dict_lengths(const uint32_t *order, const uint32_t *uniq_len, uint64_t n, uint32_t *len_plus1, hipStream_t s);
constexpr uint32_t DICT_TOTAL_BLOCKS = 1024;
// the sum of len + 1 over the n ids as up to DICT_TOTAL_BLOCKS partial sums (returns how many): one launch where only the total is wanted
uint32_t launch_dict_total(const uint32_t *uniq_len, uint64_t n, uint64_t *partial, hipStream_t s);
void launch_dict_emit(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint32_t *order, const uint64_t *uniq_pos,
                      const uint32_t *uniq_len, const uint64_t *off_excl, uint64_t n,
                      uint8_t *blob /* string area: offset 1 is blob[1] */, uint32_t *id_off, hipStream_t s,
                      const uint64_t *uniq_key = nullptr);
void launch_import_strings(const uint8_t *in, uint64_t n, const uint8_t *extra, const uint32_t *lens, const uint64_t *offs, uint64_t count,
                           uint64_t *slots, uint64_t slot_mask, uint32_t max_entries, uint64_t *uniq_pos, uint32_t *uniq_len,
                           uint32_t id_cap, uint32_t trim, uint64_t hi, P1Status *st, uint32_t *ids_out /* optional */, hipStream_t s,
                           uint64_t *uniq_key = nullptr);
void launch_export_strings(const uint8_t *in, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                           const uint32_t *order /* optional */, const uint64_t *off_excl, uint64_t n, uint8_t *bytes,
                           uint32_t *lens_out /* optional */, hipStream_t s, const uint64_t *uniq_key = nullptr);
// the reference's --mem-limit as a deterministic model (block ends)
void launch_model_sort(const uint64_t *uniq_pos, const uint32_t *uniq_len, uint32_t n, SortBuffers &b, uint32_t **vals_out, hipStream_t s);
void launch_model_trip(const uint64_t *excl, uint32_t nreal, double base_mb, double node_bytes, double limit_mb, uint64_t arena_bytes, uint64_t *out,
                       hipStream_t s);
void launch_locate_field(const uint8_t *in, uint64_t lo, uint64_t p, uint64_t *out, hipStream_t s);
// ... over the shards of a row-sharded block (zdw_shard_model_walk, zdw_owner_import / _first_flags / _finish)
void launch_model_trip_chained(const uint64_t *excl, uint32_t n, uint64_t arenas0, uint64_t entries0, uint64_t free0, double base_mb, double node_bytes,
                               double limit_mb, uint64_t arena_bytes, uint64_t *out /* 4 */, hipStream_t s);
void launch_mask_first(const uint32_t *order, const uint32_t *flags, uint32_t n, const uint32_t *uniq_len, uint32_t *mlen, uint32_t *count, hipStream_t s);
void launch_export_positions(const uint32_t *order, const uint64_t *uniq_pos, uint32_t n, uint64_t base, uint64_t *out, hipStream_t s);
void launch_owner_minpos(const uint32_t *ids, const uint64_t *pos, uint64_t n, uint64_t *minpos, hipStream_t s);
void launch_owner_first(const uint32_t *ids, const uint64_t *pos, uint64_t n, const uint64_t *minpos, uint32_t *flags, hipStream_t s);
void launch_owner_filter(const uint64_t *minpos, uint32_t *ulen, uint32_t n, uint64_t last_pos, hipStream_t s);
void launch_trim_block(const uint8_t *rec, uint32_t rec_stride, uint32_t ncols, const uint8_t *col_kind, const uint32_t *col_recoff, uint32_t rows,
                       uint32_t partial, uint8_t *live, const uint64_t *uniq_pos, uint64_t last_pos, uint64_t *col_min, uint64_t *col_max,
                       uint8_t *col_seen, uint32_t *uniq_len, uint32_t nids, uint32_t *entries, const uint32_t *nz, uint32_t nz_stride, hipStream_t s);
// range-partitioned dictionary (multi-GPU)
void launch_key16_at(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                     const uint32_t *order, const uint32_t *idx, uint32_t n, uint64_t *keys, hipStream_t s);
void launch_key16_sample(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                         const uint32_t *idx, uint32_t n, uint32_t nids, uint64_t *keys, hipStream_t s);
struct SortBuffers;
void launch_bucket_partition(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                             uint32_t nids, const uint64_t *splitters, uint32_t nparts, SortBuffers &b, uint32_t *order_out, uint32_t *bounds,
                             hipStream_t s, const uint64_t *uniq_key = nullptr);
void launch_partition_search(const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *uniq_pos, const uint32_t *uniq_len,
                             const uint32_t *order, uint32_t lo, uint32_t hi, const uint64_t *splitters, uint32_t nsplit, uint32_t *bounds,
                             hipStream_t s);
void launch_owner_offsets(const uint32_t *ids, const uint32_t *id_off, uint32_t base, uint64_t n, uint32_t *out, hipStream_t s);
void launch_scatter_offsets(const uint32_t *order, const uint32_t *offsets, uint64_t n, uint32_t *id_off, hipStream_t s);

// pass 2
void launch_pass2_size(const Pass2Args &a, hipStream_t s);
void launch_pass2_emit(const Pass2Args &a, hipStream_t s);
void launch_last_row(const Pass2Args &a, hipStream_t s);

}  // namespace zdw
