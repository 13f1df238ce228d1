/*
 * zdw_hip.h — C ABI of libzdw_hip.so: the MI355X (gfx950) implementation of adobe/zdw's
 * two-pass TSV -> ZDW block encode.
 *
 * The reference has no FFI for this path; its seam is four private calls that
 * ConvertToZDW::processFile makes once per block (cplusplus/ConvertToZDW.cpp:805,848,851,872):
 *
 *     parseInput(FILE*)                -> zdw_block_pass1
 *     uniques.write(FILE*)             -> zdw_block_dict
 *     writeLookupColumnStats(FILE*,n)  -> zdw_block_stats
 *     writeBlockRows(FILE*,FILE*,n,u)  -> zdw_block_pass2
 *     uniques.clear()                  -> zdw_block_next
 *
 * They stream through FILE*; this ABI is buffer based (plain pointers and sizes, no C++ or
 * torch types).  The host shim (zdw_amd/csrc/ConvertToZDW.cpp, the convertDWfile CLI) keeps
 * everything else: .desc.sql parsing, file header, block header, compressor child, rename.
 *
 * Return values are the reference's ERR_CODE numbers (cplusplus/ConvertToZDW.h:44-70):
 * 0 OK, 7 OUT_OF_MEMORY, 15 WRONG_NUM_OF_COLUMNS_ON_A_ROW, 20 UNKNOWN_ERROR (HIP failures,
 * unsupported input); zdw_last_error() has the text.  No exception crosses this boundary.
 * zdw_block_dict returns 7 when the block's dictionary would exceed zdw_opts.max_dict_bytes (or the
 * format's 32-bit offsets): the caller re-runs zdw_block_pass1 for the same block with a smaller
 * zdw_split_hint.max_rows (the reference closes a block for the same reason, memory, at an
 * environment-dependent point).
 *
 * Threading: one context per input file; calls on one context must be serialised.
 * Ownership: the library owns all device memory and every buffer it returns (valid until the
 * next zdw_block_next / zdw_set_input_* / zdw_ctx_destroy on that context); the caller owns
 * what it passes in.
 */
#ifndef ZDW_HIP_H
#define ZDW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZDW_ABI_VERSION 1

/* ERR_CODE values used by this library (cplusplus/ConvertToZDW.h:44-70) */
#define ZDW_OK 0
#define ZDW_OUT_OF_MEMORY 7
#define ZDW_WRONG_NUM_OF_COLUMNS_ON_A_ROW 15
#define ZDW_UNKNOWN_ERROR 20

/* column type ids stored in the file (cplusplus/zdw_column_type_constants.h:17-35) */
enum {
    ZDW_VARCHAR = 0, ZDW_TEXT = 1, ZDW_DATETIME = 2, ZDW_CHAR_2 = 3, ZDW_CHAR = 6,
    ZDW_TINY = 7, ZDW_SHORT = 8, ZDW_LONG = 9, ZDW_LONGLONG = 10, ZDW_DECIMAL = 11,
    ZDW_TINY_SIGNED = 12, ZDW_SHORT_SIGNED = 13, ZDW_LONG_SIGNED = 14, ZDW_LONGLONG_SIGNED = 15,
    ZDW_TINYTEXT = 16, ZDW_MEDIUMTEXT = 17, ZDW_LONGTEXT = 18
};

typedef struct zdw_ctx zdw_ctx;

typedef struct zdw_opts {
    uint32_t struct_size;          /* sizeof(zdw_opts), for forward compatibility */
    int32_t device;                /* HIP device ordinal; -1 = the calling thread's current device */
    uint32_t trim_trailing_spaces; /* convertDWfile -t (ConvertToZDW.cpp:295-313) */
    uint32_t first_row_capacity;   /* initial m_LongestLine; 0 = 16384 (ConvertToZDW.cpp:965) */
    uint64_t dict_slots;           /* initial hash-table slots for the string dictionary; 0 = automatic */
    uint64_t max_dict_bytes;       /* a block's dictionary may not exceed this; 0 = the format's limit (2^32 - 1) */
    uint64_t max_device_bytes;     /* device memory this context may hold; 0 = no limit of its own.  A block whose buffers do not
                                      fit (this limit or the device itself) ends earlier, as the reference ends a block when its
                                      memory limit trips (--mem-limit, stringheap.cpp:74-86) */
    uint64_t mem_limit_mb;         /* convertDWfile --mem-limit=<MB> (convertDWfile.cpp:161-163, default 3072 there; 0 here = off): the
                                      reference ends a block when its StringHeap takes a new 64 MiB arena while the process is at or
                                      beyond this virtual size (stringheap.cpp:74-86, memory.cpp:76-81).  Reproduced as a
                                      deterministic model of that size (fitted to the reference build on this image):
                                      2.6 + 64 x arenas + 64 B x dictionary entries.  A zdw_split_hint overrides it. */
    uint64_t track_first_occurrence; /* 1: for every unique string of a block pass 1 keeps the position of its FIRST occurrence in this context's
                                      image (a minimum over all of them), whichever occurrence won the insertion: what the block-limit model needs
                                      when the block is row-sharded over several contexts (zdw_shard_model_walk); implied by mem_limit_mb */
    uint64_t defer_bad_row;        /* 1: a row with a wrong field count does not fail zdw_block_pass1 (15): the call returns 0 with
                                      zdw_pass1_result.bad_row set and the caller decides -- a row-sharded block under a memory limit may end
                                      before that row, which then belongs to a later block (ConvertToZDW.cpp:336-338: the reference meets
                                      rows in order and stops at whichever comes first) */
} zdw_opts;

/* Where a block must end when it cannot hold the whole remaining input (reference: memory-driven,
 * StringHeap/Memory).  max_rows = rows in this block; partial_fields = how many
 * leading fields of the following row pass 1 had already visited when the limit tripped (they stay
 * in this block's dictionary and ranges, the row itself moves to the next block). */
typedef struct zdw_split_hint {
    uint32_t max_rows;        /* 0xFFFFFFFF = no limit */
    uint32_t partial_fields;
} zdw_split_hint;

typedef struct zdw_pass1_result {
    uint32_t num_rows;        /* rows in this block (block header field 1) */
    uint32_t longest_line;    /* emulated m_LongestLine after this block's pass 1 (block header field 2) */
    uint32_t is_last_block;   /* block header field 3 ('done') */
    uint32_t bad_row;         /* 1-based row within the block when the call returns 15, else 0 */
    uint64_t bytes_consumed;  /* input bytes covered by blocks so far (start of the next block) */
    uint64_t dict_entries;    /* unique strings this block's pass 1 found (Dictionary::getNumEntries) */
    uint64_t table_slots;     /* hash-table size that was sufficient */
} zdw_pass1_result;

/* Per-kernel device times of the last block, milliseconds, measured with HIP events on the
 * context's stream when timing is enabled (zdw_ctx_enable_timing). */
#define ZDW_NUM_STAGES 12
typedef struct zdw_stage_times {
    float ms[ZDW_NUM_STAGES];
    uint32_t launches[ZDW_NUM_STAGES];
} zdw_stage_times;
enum {
    ZDW_STAGE_SCAN_ROWS = 0,   /* row-start count per span + exclusive scan            */
    ZDW_STAGE_PASS1 = 1,       /* tokenise + hash/dedupe + integer parse + column ranges */
    ZDW_STAGE_DICT_COMPACT = 2, /* unused since unique ids are dense (kept so that stage numbers stay put) */
    ZDW_STAGE_DICT_SORT = 3,
    ZDW_STAGE_DICT_EMIT = 4,   /* offsets + dictionary blob */
    ZDW_STAGE_PASS2_SIZE = 5,  /* per-row encoded length + exclusive scan */
    ZDW_STAGE_PASS2_EMIT = 6,
    ZDW_STAGE_H2D = 7,
    ZDW_STAGE_D2H = 8
};

uint32_t zdw_abi_version(void);

/* col_types: ncols type ids as stored in the file header (after ReadDescFile, ConvertToZDW.cpp:91-162). */
int zdw_ctx_create(const uint8_t *col_types, uint32_t ncols, const zdw_opts *opts, zdw_ctx **out);
void zdw_ctx_destroy(zdw_ctx *ctx);
const char *zdw_last_error(const zdw_ctx *ctx);   /* ctx may be NULL: last zdw_ctx_create error */
int zdw_ctx_enable_timing(zdw_ctx *ctx, int enable);
int zdw_ctx_stage_times(const zdw_ctx *ctx, zdw_stage_times *out);

/* The whole TSV image of one input file.  Host variant copies to HBM (pinned staging);
 * the device variant borrows an HBM-resident buffer (16-byte aligned) and never writes to it. */
int zdw_set_input_host(zdw_ctx *ctx, const uint8_t *tsv, uint64_t len);
int zdw_set_input_device(zdw_ctx *ctx, const void *d_tsv, uint64_t len);

/* The input as a stream (the reference reads its FILE* row by row, and with -i tees stdin into a temporary file that it rewinds for
 * pass 2: ConvertToZDW.cpp:786-803,854-868, convertDWfile.cpp:144): `read` is called from a thread of the library, in order, with a
 * buffer of host memory to fill; it returns the bytes it delivered (fewer than cap only at the end of the input), 0 at the end and a
 * negative number on failure (reported as 20 by the next zdw_block_pass1).  The bytes travel to the device in pieces while earlier
 * blocks are encoded; no host copy of the input is kept (the device image is: pass 2 and the dictionary read it again).
 * zdw_block_pass1 waits until the rows its block needs have arrived -- all of the input for a block without a limit, its row
 * window for a block under zdw_opts.mem_limit_mb or a split hint.  size_hint = the input's length when it is known (a file), else 0
 * (the device image then grows by doubling).  The call returns at once.  zdw_ctx_destroy, or setting another input, waits for a
 * `read` in progress to return and makes no further call. */
typedef int64_t (*zdw_read_fn)(void *user, uint8_t *buf, uint64_t cap);
int zdw_set_input_stream(zdw_ctx *ctx, zdw_read_fn read, void *user, uint64_t size_hint);
/* The same for an input that already lies in host memory (a mapped file): the library's thread copies it to the device from where
 * it lies, piece by piece, while earlier blocks are encoded; the call returns at once and `tsv` must stay valid until the context
 * is destroyed or given another input.  (zdw_set_input_host copies everything before it returns.) */
int zdw_set_input_host_async(zdw_ctx *ctx, const uint8_t *tsv, uint64_t len);

/* parseInput(): consumes rows from the current position up to the end of input or the split hint
 * (NULL = no limit).  Returns 0, 15 (res->bad_row set), 7 (a block of zero rows was forced) or 20. */
int zdw_block_pass1(zdw_ctx *ctx, const zdw_split_hint *hint, zdw_pass1_result *res);
/* pass-1 kernel launches of this context so far (a block that the memory-limit model ends takes one over a window of rows a little larger
 * than the block -- what lies beyond its end is taken back out of the records --, another one when the window was too short) */
uint64_t zdw_ctx_pass1_launches(const zdw_ctx *ctx);
/* leading fields of the row after the current block that its pass 1 visited (zdw_split_hint.partial_fields as taken: what the caller
 * passed, or what the memory-limit model found; 0 for a block that ends the input or that device memory ended) */
uint32_t zdw_block_partial_fields(const zdw_ctx *ctx);

/* Dictionary::write(): the dictionary section exactly as it appears in the file
 * (dictionary.cpp:76-111): 00 for an empty dictionary, else index_size, size (index_size bytes LE),
 * 00, then every string + NUL in strcmp order. */
int zdw_block_dict(zdw_ctx *ctx, const uint8_t **section, uint64_t *section_len, uint8_t *index_size);

/* writeLookupColumnStats(): col_size[ncols]; col_base[used] for used columns in column order
 * (ConvertToZDW.cpp:417-483).  col_base must have room for ncols entries. */
int zdw_block_stats(zdw_ctx *ctx, uint8_t *col_size, uint64_t *col_base, uint32_t *used);

/* writeBlockRows(): the encoded rows of the block (ConvertToZDW.cpp:486-606). */
int zdw_block_pass2(zdw_ctx *ctx, const uint8_t **rows, uint64_t *rows_len);

/* Device-resident forms of the two bulk outputs (no D2H copy): for callers that keep the block in HBM. */
int zdw_block_dict_device(zdw_ctx *ctx, const void **d_section, uint64_t *section_len, uint8_t *index_size);
int zdw_block_pass2_device(zdw_ctx *ctx, const void **d_rows, uint64_t *rows_len);

/* The same two outputs handed to a writer piece by piece (the reference writes them straight into the compressor's
 * pipe, dictionary.cpp:76-111 and ConvertToZDW.cpp:486-606): `write` is called in order with consecutive pieces
 * (host memory, valid during the call) and returns 0 to go on; while it runs, the next piece is already on its way
 * from the device.  No host buffer of the output's size is needed.  A non-zero return of `write` stops the
 * transfer and is returned as ZDW_UNKNOWN_ERROR (20). */
typedef int (*zdw_write_fn)(void *user, const uint8_t *bytes, uint64_t len);
int zdw_block_dict_write(zdw_ctx *ctx, zdw_write_fn write, void *user, uint64_t *section_len, uint8_t *index_size);
int zdw_block_pass2_write(zdw_ctx *ctx, zdw_write_fn write, void *user, uint64_t *rows_len);

/* uniques.clear(): drop the block's state; the next zdw_block_pass1 continues at bytes_consumed. */
int zdw_block_next(zdw_ctx *ctx);

/* ---- multi-GPU: one ZDW block row-sharded over several contexts ----------------
 * After zdw_block_pass1 on every shard:
 *   zdw_shard_export_ranges / zdw_shard_import_ranges  : per-column min / max / seen (all-reduce)
 *   zdw_shard_export_uniques                           : this shard's unique strings (len-prefixed)
 *   zdw_shard_import_uniques                           : the union of all shards' strings; every
 *                                                        shard then sorts the same set -> same offsets
 *   zdw_shard_set_prev_row / zdw_shard_last_row        : stored values of the row before this shard
 */
int zdw_shard_export_ranges(zdw_ctx *ctx, uint64_t *col_min, uint64_t *col_max, uint8_t *col_seen);
int zdw_shard_import_ranges(zdw_ctx *ctx, const uint64_t *col_min, const uint64_t *col_max, const uint8_t *col_seen);
int zdw_shard_export_uniques(zdw_ctx *ctx, const void **d_lens, const void **d_bytes, uint64_t *count, uint64_t *nbytes);
int zdw_shard_import_uniques(zdw_ctx *ctx, const void *d_lens, const void *d_bytes, uint64_t count, uint64_t nbytes);
int zdw_shard_last_row(zdw_ctx *ctx, uint64_t *stored /* ncols */);
/* m_LongestLine of a row-sharded block (block header field 2).  The reference's line buffer starts at 16384, doubles
 * while a row does not fit and never shrinks (getnextrow.cpp:57-65), so what it holds after a shard's rows depends on
 * what it held before them.  caps[k] = the buffer after this shard's rows when it held 16384 << k before them
 * (k = 0 .. ZDW_LINE_CAPS - 1): composing the shards' tables in row order gives the block's value without a
 * sequential chain of exchanges.  zdw_pass1_result.longest_line is the k = 0 entry of a context's first block. */
#define ZDW_LINE_CAPS 18
int zdw_shard_line_caps(zdw_ctx *ctx, uint32_t *caps /* ZDW_LINE_CAPS */);
int zdw_shard_set_prev_row(zdw_ctx *ctx, const uint64_t *stored /* ncols, NULL = zeros */);

/* Range-partitioned dictionary (instead of export / import of the union): the node-wide dictionary is sorted
 * once, 1/N of it on every rank.  Per shard, after zdw_block_pass1 and the range exchange:
 *   zdw_shard_sort_uniques    take stock of this shard's unique strings (count = how many).  Despite the name a
 *                             shard does not sort them (ZDW_SHARD_PRESORT=1 brings the sorting form back)
 *   zdw_shard_sample_keys     nsamples 16-byte prefix keys of its strings (2 big-endian u64 each; all ones = no
 *                             string) -> the caller gathers all shards' samples and picks N-1 splitters (the
 *                             same on every shard)
 *   zdw_shard_partition       strings grouped by part as lens + NUL-terminated bytes; part p = keys in
 *                             [splitter[p-1], splitter[p]); counts / nbytes per part -> all-to-all
 *   zdw_owner_build           (as owner of range r) dedupe + sort + lay out the strings received for it
 *   zdw_owner_offsets         dictionary offset of every received string, in arrival order -> all-to-all back
 *   zdw_shard_set_offsets     offsets of this shard's strings (in the order they were sent); from then on
 *                             zdw_block_dict[_device] returns the piece of the section this rank owns
 *                             (ranks 0..N-1 concatenated = the reference's section). */
int zdw_shard_sort_uniques(zdw_ctx *ctx, uint64_t *count);
int zdw_shard_sample_keys(zdw_ctx *ctx, uint32_t nsamples, uint64_t *keys /* 2 * nsamples */);
int zdw_shard_partition(zdw_ctx *ctx, uint32_t nparts, const uint64_t *splitters /* 2 * (nparts - 1) */, uint64_t *counts /* nparts */,
                        uint64_t *nbytes /* nparts */, const void **d_lens, const void **d_bytes);
int zdw_owner_build(zdw_ctx *ctx, const void *d_lens, const void *d_bytes, uint64_t count, uint64_t nbytes, uint64_t *slice_bytes);
int zdw_owner_offsets(zdw_ctx *ctx, uint64_t base /* dictionary offset of the range's first string */, const void **d_offsets);
int zdw_shard_set_offsets(zdw_ctx *ctx, const void *d_offsets, uint64_t count, uint64_t dict_size, int first_range);
/* the owner's half of zdw_shard_set_offsets alone, for a context that owns a key range of the block's dictionary but holds none of
 * its rows (a shard whose rows all lie in earlier or later blocks): zdw_block_dict[_device|_write] then returns its piece */
int zdw_owner_section(zdw_ctx *ctx, uint64_t dict_size, int first_range);

/* ---- where the reference's memory limit ends a row-sharded block (SURVEY.md §8e "Invariance requirement", Q9) ------------------
 * The reference ends a block when its StringHeap takes a new arena while the process is at or beyond --mem-limit
 * (stringheap.cpp:74-86, memory.cpp:76-81, ConvertToZDW.cpp:338,404): a function of the order in which the block's strings are FIRST
 * met.  Over row shards that order is: shard 0's strings in the order of their first occurrence, then those of shard 1 that shard 0
 * does not hold, and so on -- so the decision is taken on the node-wide dictionary, never per shard:
 *   contexts are created with zdw_opts.track_first_occurrence (and mem_limit_mb = 0: no shard ends a block on its own);
 *   zdw_shard_model_counts   entries / bytes / longest string of this shard: with zdw_model_upper_bound_mb the caller sees cheaply
 *                            that a limit cannot trip (all shards' counts added up as if no string were shared);
 *   zdw_shard_positions      after zdw_shard_partition: the position of every exported string's first occurrence, `base` (where this
 *                            shard's image begins in the whole input) added -- travels to the owners with the strings;
 *   zdw_owner_import         (owner) the first half of zdw_owner_build: the arriving strings deduplicated, the smallest position kept;
 *   zdw_owner_first_flags    (owner) per arrival: 1 = this is the string's first occurrence node-wide -> back to the requesters;
 *   zdw_shard_model_walk     (requester, shards in row order, the state handed from one to the next) the reference's arenas walked over
 *                            this shard's node-wide-first strings in the order of their positions: *found = the limit trips in this shard,
 *                            at field *fields - 1 (0-based) of its row *row, which starts at *row_start; *trip_pos = where that field lies;
 *   zdw_shard_trim           the shard that holds the end: its block is cut back to `row` rows and the first `fields` fields of the next
 *                            row (they stay in the block's dictionary and ranges as in the reference); the shards behind it take no part
 *                            in this block; zdw_block_next then moves the shard on to row_start;
 *   zdw_owner_finish         (owner) the second half of zdw_owner_build: strings first met beyond base + trip_pos are dropped
 *                            (last_pos; ~0 = keep all), the rest sorted and laid out.
 * zdw_owner_build = zdw_owner_import without positions + zdw_owner_finish(~0). */
typedef struct zdw_model_state {
    uint64_t arenas;       /* arenas the block's strings have taken so far */
    uint64_t entries;      /* dictionary entries so far */
    uint64_t arena_free;   /* bytes left in the arena that is open */
} zdw_model_state;
double zdw_model_upper_bound_mb(uint64_t entries, uint64_t bytes, uint32_t max_len);
int zdw_shard_model_counts(zdw_ctx *ctx, uint64_t *entries, uint64_t *bytes, uint32_t *max_len);
int zdw_shard_positions(zdw_ctx *ctx, uint64_t base, const void **d_pos);
int zdw_owner_import(zdw_ctx *ctx, const void *d_lens, const void *d_bytes, const void *d_pos /* NULL: no positions */, uint64_t count, uint64_t nbytes);
int zdw_owner_first_flags(zdw_ctx *ctx, const void **d_flags);
int zdw_owner_finish(zdw_ctx *ctx, uint64_t last_pos, uint64_t *slice_bytes);
/* Row lengths beside the owners' sort.  A text column's width is the byte width of the dictionary's SIZE, not of its order:
 *   zdw_owner_size           (owner, after zdw_owner_import) the bytes its range will take (sum of len + 1 of its deduplicated strings);
 *   zdw_shard_rows_early     (requester) told the whole dictionary's size (1 + the owners' figures), starts its row lengths on the
 *                            context's second stream, as if no row preceded its first; zdw_shard_set_prev_row has that row done again.
 * Both are optional: without them the row lengths are computed when the rows are asked for (ConvertToZDW.cpp:486-606 has no such order). */
int zdw_owner_size(zdw_ctx *ctx, uint64_t *bytes);
int zdw_shard_rows_early(zdw_ctx *ctx, uint64_t dict_size);
int zdw_shard_model_walk(zdw_ctx *ctx, const void *d_flags, uint64_t count, double limit_mb, zdw_model_state *state, uint32_t *found,
                         uint32_t *row, uint32_t *fields, uint64_t *row_start, uint64_t *trip_pos);
int zdw_shard_trim(zdw_ctx *ctx, uint32_t row, uint32_t fields, uint64_t row_start, zdw_pass1_result *res);

/* ---- the same exchange inside the library: one process, several GPUs of one node -----------------
 * One ZDW block row-sharded over `ndev` devices: a zdw_ctx and a host thread per device, column ranges by
 * ncclAllReduce, the range-partitioned dictionary by grouped ncclSend / ncclRecv over the xGMI links (RCCL,
 * ncclCommInitAll: no second process, nothing re-executed), the row halo by the host.  The bytes are the ones a single
 * context writes for the whole input, whatever ndev is.  A device may be listed more than once (tests on a one-GPU
 * box): the exchange then runs as device-to-device copies.  convertDWfile --gpus=N (not in the reference tool) uses it.
 * Without zdw_opts.mem_limit_mb the whole input is ONE block (a dictionary beyond 2^32 bytes returns 7); a shard that does not fit its
 * device returns 7. */
typedef struct zdw_multi zdw_multi;
typedef struct zdw_multi_block {
    uint32_t num_rows;        /* block header fields 1-3 (ConvertToZDW.cpp:839-842) */
    uint32_t longest_line;
    uint32_t is_last_block;
    uint32_t bad_row;         /* 1-based row within the block when zdw_multi_encode returns 15, else 0 */
    uint64_t dict_bytes;      /* Dictionary::getSize() of the whole block */
} zdw_multi_block;
int zdw_multi_create(const uint8_t *col_types, uint32_t ncols, const zdw_opts *opts, const int32_t *devices, uint32_t ndev, zdw_multi **out);
void zdw_multi_destroy(zdw_multi *m);
const char *zdw_multi_last_error(const zdw_multi *m);   /* m may be NULL: last zdw_multi_create error */
/* cuts the image into ndev pieces at row boundaries and copies piece i to device i; tsv must stay valid until zdw_multi_encode returns */
int zdw_multi_set_input_host(zdw_multi *m, const uint8_t *tsv, uint64_t len);
/* the same for pieces that already lie on the devices: piece i (whole rows, 16-byte aligned, lens[i] bytes) on device i, in row order;
 * nothing is copied and the buffers must stay valid until the next input or zdw_multi_destroy */
int zdw_multi_set_input_device(zdw_multi *m, const void *const *d_pieces, const uint64_t *lens);
/* the next block of the input.  With zdw_opts.mem_limit_mb the input's blocks end where the reference's end (the model of its memory
 * limit walked over the node-wide dictionary, see zdw_shard_model_walk); is_last_block = 0 then says that zdw_multi_next + another
 * zdw_multi_encode give the block behind it.  num_rows = 0 with is_last_block = 1: the input holds no (more) rows. */
int zdw_multi_encode(zdw_multi *m, zdw_multi_block *blk);
/* uniques.clear() for the whole set of shards: on to the block behind the one zdw_multi_encode returned */
int zdw_multi_next(zdw_multi *m);
int zdw_multi_stats(zdw_multi *m, uint8_t *col_size, uint64_t *col_base, uint32_t *used);
/* dictionary section / rows of the block, shard after shard, through the writer (see zdw_block_dict_write) */
int zdw_multi_dict_write(zdw_multi *m, zdw_write_fn write, void *user, uint64_t *section_len);
int zdw_multi_rows_write(zdw_multi *m, zdw_write_fn write, void *user, uint64_t *rows_len);

/* ---- decode: a raw ZDW stream back to its TSV rows -----------------------------------------------
 * The other direction of the format, block by block, on the device: what UnconvertFromZDW does in readHeader / readNextBlock /
 * readNextRow (cplusplus/UnconvertFromZDW.cpp:758-1000, 1270-1462; versions 9-11, the column types the encoder writes).  The rows
 * come out as the reference prints them: tab-separated, one line per row, an unused column as its default ("" for text-like
 * columns and char(1), 0 for integers, 0.000000000000 for decimal).  Return values are the reference decoder's ERR_CODE numbers
 * (cplusplus/zdw/UnconvertFromZDW.h:34-56): 0 OK, 5 unsupported version, 6 the data ends before the block does, 7 unexpected
 * column type, 9 corrupted data; 20 for a device error (zdw_dec_last_error has the text). */
typedef struct zdw_dec zdw_dec;
typedef struct zdw_dec_block {
    uint32_t num_rows;
    uint32_t longest_line;
    uint32_t is_last_block;
    uint32_t reserved;
    uint64_t text_bytes;      /* length of the block's rows as text */
    uint64_t zdw_bytes;       /* bytes of the stream this block took, its 9-byte header included */
    uint64_t dict_bytes;
    uint64_t walk_passes;     /* how often the row chain had to be walked again from corrected entries (1 = the guesses were right) */
} zdw_dec_block;
int zdw_dec_create(int device, zdw_dec **out);
void zdw_dec_destroy(zdw_dec *d);
const char *zdw_dec_last_error(const zdw_dec *d);      /* d may be NULL: last zdw_dec_create error */
/* the whole stream (uncompressed .zdw) in host memory: copied to the device, file header parsed */
int zdw_dec_set_input_host(zdw_dec *d, const uint8_t *zdw, uint64_t len);
int zdw_dec_header(const zdw_dec *d, uint32_t *version, uint32_t *ncols);
/* column i: name (NUL-terminated, valid until the next input), type id, char size field */
int zdw_dec_column(const zdw_dec *d, uint32_t i, const char **name, uint8_t *type, uint16_t *char_size);
/* decodes the next block on the device; returns 0 and blk->num_rows == 0 with is_last_block set when the stream holds no (more) blocks */
int zdw_dec_next_block(zdw_dec *d, zdw_dec_block *blk);
/* the block's rows as text: left on the device, or handed to a writer in pieces (see zdw_block_dict_write) */
int zdw_dec_text_device(zdw_dec *d, const void **d_text, uint64_t *text_len);
int zdw_dec_text_write(zdw_dec *d, zdw_write_fn write, void *user, uint64_t *text_len);
/* milliseconds of device work of the last block by stage: 0 row chain, 1 expand, 2 repeat-fill, 3 text lengths + offsets, 4 text */
int zdw_dec_stage_ms(const zdw_dec *d, float ms[5]);

#ifdef __cplusplus
}
#endif
#endif /* ZDW_HIP_H */
