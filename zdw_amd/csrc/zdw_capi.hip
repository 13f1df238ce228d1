// zdw_capi.hip — the C ABI of libzdw_hip.so (include/zdw_hip.h): host orchestration of the
// gfx950 kernels for one ZDW block at a time.  Mirrors the per-block seam of
// ConvertToZDW::processFile (cplusplus/ConvertToZDW.cpp:765-894).
#include "../../include/zdw_hip.h"
#include "zdw_knobs.h"
#include "zdw_dev.h"
#include "zdw_kernels.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

using namespace zdw;

namespace {

thread_local std::string g_create_error;

// Device memory a context may hold (zdw_opts.max_device_bytes; 0 = whatever the device gives): lets tests and
// callers that share a GPU make the encoder end its blocks early instead of failing.
struct Budget { size_t used = 0, cap = 0; };

static double g_alloc_ms = 0.0;      // time inside hipMalloc / hipFree (ZDW_DEBUG_RETRY prints it)
struct Lap {      // ZDW_DEBUG_LAPS=1: wall time between marks of one call, on stderr
    const bool on = experiment("ZDW_DEBUG_LAPS") != nullptr;
    double t0 = 0, last = 0;
    std::string line;
    void mark(const char *what);
    void done(const char *name);
};
static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void Lap::mark(const char *what)
{
    if (!on) return;
    const double t = now_ms();
    if (t0 == 0) { t0 = last = t; return; }
    char b[64];
    snprintf(b, sizeof b, " %s=%.1f", what, t - last);
    line += b;
    last = t;
}
void Lap::done(const char *name) { if (on) fprintf(stderr, "[laps] %s:%s total=%.1f ms\n", name, line.c_str(), now_ms() - t0); }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    Budget *bud = nullptr;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap && p) return hipSuccess;
        size_t want = bytes < 256 ? 256 : bytes;
        // Sizes that follow the number of unique ids jitter by a few hundred from pass to pass (which wave wins which
        // race), and every regrowth is a hipFree (a device synchronisation) plus a hipMalloc in the middle of a block:
        // a large buffer gets 3 % more than asked, one that has to grow an eighth more.
        if (want >= (1u << 20)) want += want / 32;
        if (p) { want += want / 8; release(); }
        if (bud && bud->cap && bud->used + want > bud->cap) return hipErrorOutOfMemory;
        const double t0 = now_ms();
        hipError_t e = hipMalloc(&p, want);
        g_alloc_ms += now_ms() - t0;
        if (e == hipSuccess) { cap = want; if (bud) bud->used += want; }
        else { p = nullptr; (void)hipGetLastError(); }      // handled here: do not leave the error behind for a later hipGetLastError()
        return e;
    }
    void release()
    {
        const double t0 = now_ms();
        if (p) { (void)hipFree(p); if (bud) bud->used -= cap; }
        g_alloc_ms += now_ms() - t0;
        p = nullptr; cap = 0;
    }
    template <typename T> T *as() const { return static_cast<T *>(p); }
};

struct HostBuf {   // pinned
    void *p = nullptr;
    size_t cap = 0;
    hipError_t ensure(size_t bytes)
    {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { (void)hipHostFree(p); p = nullptr; cap = 0; }
        size_t want = bytes < 4096 ? 4096 : bytes;
        hipError_t e = hipHostMalloc(&p, want, hipHostMallocDefault);
        if (e == hipSuccess) cap = want; else p = nullptr;
        return e;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; cap = 0; }
};

unsigned bytes_for(uint64_t v)   // dictionary.cpp:62-73 and ConvertToZDW.cpp:458-464
{
    unsigned n = 1;
    while (v >= 256) { ++n; v /= 256; }
    return n;
}

struct TimedSpan { hipEvent_t a, b; int stage; };

}  // namespace

// ---- an input that arrives through a read callback while earlier blocks are encoded (zdw_set_input_stream) ----
// A feeder thread reads pieces into pinned buffers and copies them to the device image on its own stream; the image of a stream
// of unknown length grows by doubling (a device-to-device copy; the smaller image lives on until the block that may be reading
// it is done).  Pass 1 takes a snapshot (pointer, bytes that have arrived, end seen?) and waits when the block needs more.
struct InputStream {
    zdw_read_fn read = nullptr;
    void *user = nullptr;
    const uint8_t *host = nullptr;         // instead of `read`: the input lies in host memory (a mapped file), size_hint bytes of it
    int device = 0;
    uint64_t size_hint = 0, piece_bytes = 32ull << 20, first_cap = 256ull << 20;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    // under m:
    void *img = nullptr;
    uint64_t cap = 0, uploaded = 0;
    bool final = false, stop = false;
    int error = 0;                         // 1: the callback failed, 2: a device call failed
    std::string error_text;
    std::vector<void *> retired;           // earlier images: freed when the context moves on to its next block
};

struct zdw_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;         // pass 2's row lengths run here while the dictionary is sorted on `stream`
    std::string err;
    // schema
    uint32_t ncols = 0, nnum = 0, rec_stride = 0;
    std::vector<uint8_t> types, kind;
    std::vector<uint32_t> recoff;
    std::vector<int32_t> numidx;
    DevBuf d_kind, d_recoff, d_numidx;
    bool trim = false;
    uint32_t row_cap = FIRST_ROW_CAP;      // emulated m_LongestLine, persists across the blocks of one input
    uint32_t first_row_cap = FIRST_ROW_CAP; // what it starts from for every input (ConvertToZDW.cpp:965)
    uint32_t row_cap_block = FIRST_ROW_CAP; // its value when the current block began (a re-run of the block starts from it again)
    uint64_t row_cap_block_lo = ~0ull;
    Budget budget;
    double mem_limit_mb = 0.0;             // the reference's --mem-limit as a model of where its blocks end (0 = off)
    uint64_t model_max_slots = 0;          // while the model drives pass 1: the dictionary table never grows beyond this
    uint32_t model_rows_prev = 0;          // rows of the previous block the model ended (the next block's first window)
    uint32_t model_rows_prev2 = 0;         // ... and of the one before it
    double model_vsize_end = 0.0;          // the model's upper bound of the process size at the end of the rows pass 1 last saw (MB)
    uint32_t partial_taken = 0;            // leading fields of the row after the current block that its pass 1 visited (zdw_block_partial_fields)
    uint64_t pass1_launches = 0;           // pass-1 kernel launches of this context (zdw_ctx_pass1_launches)
    uint32_t last_span_shift = 14;         // geometry of the last pass 1 (model_trip)
    bool last_fold = false;                // ... which counted its own rows (span_state instead of span_rowbase)
    uint64_t last_span0 = 0;
    bool oom_seen = false;                 // a call ended in OUT_OF_MEMORY: the next pass 1 starts from released block buffers
    std::vector<LongRow> long_list;        // this block's rows of 16383 bytes or more, by position (zdw_shard_line_caps replays them)
    uint64_t long_read_end = 0;
    bool long_tail = false;
    uint64_t slots_opt = 0, slots_learned = 0, max_dict_bytes = 0;
    uint32_t rows_learned = 0;             // rows of the block slots_learned was enough for
    double entries_per_row = 0.0;          // dictionary entries per row of the last pass 1 that completed
    // input image
    const uint8_t *d_in = nullptr;
    uint64_t n = 0;
    DevBuf in_owned;
    bool have_input = false;
    InputStream *in_stream = nullptr;      // the input arrives through a callback (zdw_set_input_stream)
    bool input_open = false;               // ... and its end has not been seen yet: the image is a prefix, its last row may be incomplete
    uint64_t hi = 0;                       // effective end: last row terminator + 1
    uint64_t lo = 0;                       // start of the current block
    uint64_t next_lo = 0;                  // start of the next block (set by pass 1)
    // block state
    bool p1_done = false, dict_size_done = false, dict_done = false, stats_done = false, p2_size_done = false, p2_done = false, is_last = true;
    // a shard's row lengths started early (zdw_shard_rows_early): computed as if no row preceded its first one, for the dictionary size
    // p2_early_size; the first row is done again once the row before it is known (zdw_shard_set_prev_row)
    bool p2_size_early = false, p2_row0_stale = false;
    uint64_t p2_early_size = 0;
    uint32_t dict_max_len = 24;
    uint32_t nrows = 0;
    uint64_t nslots = 0;
    DevBuf span_starts, span_rowbase, scan_tmp, scan_tmp2, rec, slots, deferred_q, col_min, col_max, col_seen, status, long_rows, d_scalar;
    std::vector<uint64_t> h_min, h_max;
    std::vector<uint8_t> h_seen;
    // dictionary
    P1Status st_host{};           // the device's status block as the host last read it, behind the last kernel that writes it
    bool st_known = false;        // ... (pass 1, a block trim, a dictionary import): ensure_dict_size need not read it again
    DevBuf uniq_key;              // the strings' first 24 bytes by id (Pass1Args.uniq_key); left out when memory is short
    DevBuf uniq_pos, uniq_len, key_a, key_b, val_a, val_b, order, group, active_a, active_b, flags, hist, lenp1, off_excl,
        slot_off, dict_sec, extra, extra_lens, extra_offs, exp_bytes;
    uint64_t nuniq = 0, dict_size = 0, extra_bytes = 0;
    uint64_t avg_row = 0;         // bytes per row, sampled at the start of the input
    bool skip_empty = false;      // ... and: most of its fields are empty
    double bytes_per_row = 0.0;   // of the last block pass 1 completed (how many rows the next one may hold)
    DevBuf nz, d_used_col;        // tables of mostly empty fields: where pass 1 left a non-zero record (Pass1Args.nz)
    bool nz_mode = false;         // ... this block's record matrix means something only there
    bool rec_dense = true;        // ... until the rest has been zeroed for a reader that wants all of it (ensure_dense_rec)
    uint32_t nz_stride = 0;       // 32-bit words per row of the map (a bit per column; a multiple of 4 words)
    uint32_t nrows_written = 0;   // rows of the matrix pass 1 may have written (the block may have been cut back since)
    bool no_fold = false;         // pass 1 counts its own rows unless that went wrong for this input
    bool no_rows_form = false;    // the lane-per-row form's row index gave up its look-back once (cannot happen): the field form from then on
    DevBuf span_state;
    DevBuf rowidx, rx_bsmap;               // pass 1 with a lane per row: the row index (start and terminator of every row), backslashes per 16 KiB span
    bool dense_boundaries = false; // ... and whether fields are so short that pass 1 takes its large descriptor ring
    uint32_t nids = 0;            // unique ids handed out in this block (holes included): entries of id_off
    uint32_t id_cap = 0;                   // capacity of uniq_pos / uniq_len (unique ids, holes included)
    // range-partitioned dictionary (multi-GPU): this shard as a requester ...
    DevBuf my_order, part_idx, part_keys, part_bounds, part_split, part_lens;
    uint64_t sorted_n = 0, sorted_holes = 0;
    bool uniques_sorted = false, presorted = false;
    // ... and as the owner of one key range of the node-wide dictionary
    DevBuf own_key;               // ... of the owner's table (optional, like uniq_key)
    DevBuf own_slots, own_upos, own_ulen, own_status, own_extra, own_lens, own_offs, own_ids, own_idoff, own_blob, own_out;
    uint64_t own_count = 0, own_slice = 0, own_nids = 0;
    uint32_t own_max_len = 24;
    bool own_built = false, own_imported = false, own_has_pos = false;
    DevBuf own_pos, own_minpos, own_flags;  // ... where a memory limit may end the block: every arrival's first position, the smallest per string
    DevBuf model_len, exp_pos;             // requester side of that: lengths as the arena walk sees them, positions of the exported strings
    bool track_first_opt = false;          // zdw_opts.track_first_occurrence
    bool last_track = false;               // the last pass 1 kept every string's first position (Pass1Args.track_first)
    bool defer_bad_row = false;            // zdw_opts.defer_bad_row
    bool p1_all_in_one = false;            // the deferred-field queue would outgrow its budget: use the all-in-one kernel
    uint64_t dq_learned = 0;               // queue entries an earlier pass of this context needed
    double dq_rate = 0.0;                  // ... and the most it queued per input byte
    uint8_t index_size = 0;
    uint64_t dict_sec_off = 0, dict_sec_len = 0;
    uint32_t sort_rounds = 0;
    // stats
    std::vector<uint8_t> col_size;
    std::vector<uint64_t> col_base;
    std::vector<uint32_t> used;
    // pass 2
    DevBuf d_used_recoff, d_used_kind, d_used_size, d_used_base, d_prev0, d_last, row_len, row_off, rows_out, batch_pre;
    std::vector<uint64_t> prev_row;        // per column, for a shard that does not start the block
    bool have_prev = false;
    uint64_t rows_len = 0;
    HostBuf h_dict, h_rows;
    HostBuf h_piece[2];                    // zdw_block_*_write: two pinned pieces, one being written while the other arrives
    hipEvent_t piece_ev[2] = {nullptr, nullptr};
    // timing
    bool timing = false;
    std::vector<TimedSpan> spans;
    zdw_stage_times times{};
};

namespace {

int fail(zdw_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) { c->err = buf; if (code == ZDW_OUT_OF_MEMORY) c->oom_seen = true; } else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, call)                                                                                   \
    do {                                                                                                   \
        hipError_t e_ = (call);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            return fail((c), e_ == hipErrorOutOfMemory ? ZDW_OUT_OF_MEMORY : ZDW_UNKNOWN_ERROR, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                            \
    } while (0)

struct StageScope {
    zdw_ctx *c; int stage; hipEvent_t a = nullptr; hipStream_t st;
    StageScope(zdw_ctx *ctx, int stg, hipStream_t on = nullptr) : c(ctx), stage(stg), st(on ? on : ctx->stream)
    {
        if (c->timing && hipEventCreate(&a) == hipSuccess) (void)hipEventRecord(a, st);
    }
    ~StageScope()
    {
        if (!a) return;
        hipEvent_t b = nullptr;
        if (hipEventCreate(&b) == hipSuccess) {
            (void)hipEventRecord(b, st);
            c->spans.push_back({a, b, stage});
        }
    }
};

void collect_times(zdw_ctx *c)
{
    for (auto &s : c->spans) {
        float ms = 0;
        if (hipEventSynchronize(s.b) == hipSuccess && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
            c->times.ms[s.stage] += ms;
            c->times.launches[s.stage] += 1;
        }
        (void)hipEventDestroy(s.a);
        (void)hipEventDestroy(s.b);
    }
    c->spans.clear();
}

void reset_block(zdw_ctx *c)
{
    c->st_known = false;
    c->p1_done = c->dict_size_done = c->dict_done = c->stats_done = c->p2_size_done = c->p2_done = false;
    c->p2_size_early = c->p2_row0_stale = false;
    c->uniques_sorted = c->own_built = c->own_imported = false;
    c->nrows = 0; c->nuniq = 0; c->nids = 0; c->dict_size = 0; c->rows_len = 0; c->extra_bytes = 0;
    c->have_prev = false;
    c->nz_mode = false; c->rec_dense = true;
}

// Replays the reference's fgets() chunking on one logical row (getnextrow.cpp:26-84) to learn how
// far the row buffer would have grown.  `terminated` = the row ends with its unescaped newline.
void replay_row_reads(const uint8_t *row, uint64_t len, bool terminated, uint32_t *cap)
{
    auto chunk = [&](uint64_t from, uint64_t size) {   // bytes one fgets(.., size, ..) delivers
        uint64_t m = 0;
        while (m + 1 < size && from + m < len) { if (row[from + m++] == '\n') break; }
        return m;
    };
    auto ends_row = [&](uint64_t l) {
        if (row[l - 1] != '\n') return false;
        uint64_t k = 0;
        while (k + 2 <= l && row[l - 2 - k] == '\\') ++k;
        return (k & 1) == 0;
    };
    uint64_t l = chunk(0, *cap);
    if (l < 2) return;
    while (!ends_row(l)) {
        if (l == (uint64_t)*cap - 1) *cap *= 2;
        if (l >= len) return;                        // fgets hits EOF (only for an unterminated tail)
        l += chunk(l, (uint64_t)*cap - l);
    }
    (void)terminated;
}

// The line buffer after this block's rows, starting from *cap (rows that fit the running buffer cannot grow it and
// are not fetched; every fetched row grows it, so a call copies a handful of rows at most).
int replay_long_rows(zdw_ctx *c, uint32_t *cap)
{
    std::vector<uint8_t> rowbuf;
    for (const LongRow &r : c->long_list) {
        if (r.start > c->long_read_end) continue;              // belongs to a later block
        if (r.len <= (uint64_t)*cap - 1) continue;             // fits the current buffer: no growth possible
        rowbuf.resize(r.len);
        HIP_TRY(c, hipMemcpy(rowbuf.data(), c->d_in + r.start, r.len, hipMemcpyDeviceToHost));
        replay_row_reads(rowbuf.data(), r.len, true, cap);
    }
    if (c->long_tail && c->n - c->hi >= (uint64_t)*cap - 1) {
        const uint64_t tl = c->n - c->hi;
        rowbuf.resize(tl);
        HIP_TRY(c, hipMemcpy(rowbuf.data(), c->d_in + c->hi, tl, hipMemcpyDeviceToHost));
        replay_row_reads(rowbuf.data(), tl, false, cap);
    }
    return ZDW_OK;
}

int upload_schema(zdw_ctx *c)
{
    HIP_TRY(c, c->d_kind.ensure(c->ncols));
    HIP_TRY(c, c->d_recoff.ensure(c->ncols * sizeof(uint32_t)));
    HIP_TRY(c, c->d_numidx.ensure(c->ncols * sizeof(int32_t)));
    if (c->ncols) {
        HIP_TRY(c, hipMemcpy(c->d_kind.p, c->kind.data(), c->ncols, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_recoff.p, c->recoff.data(), c->ncols * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(c->d_numidx.p, c->numidx.data(), c->ncols * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    return ZDW_OK;
}

// average row length and boundary density from the image's first 64 KiB (span size, descriptor ring: zdw_block_pass1)
int sample_input(zdw_ctx *c)
{
    c->avg_row = 0;
    c->dense_boundaries = false;
    c->skip_empty = false;
    if (!c->n) return ZDW_OK;
    uint8_t sample[1 << 16];
    const uint64_t take = std::min<uint64_t>(sizeof sample, c->n);
    HIP_TRY(c, hipMemcpyAsync(sample, c->d_in, take, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    uint64_t lines = 0, tabs = 0, empties = 0;
    for (uint64_t i = 0; i < take; ++i) {
        lines += sample[i] == '\n'; tabs += sample[i] == '\t';
        empties += sample[i] == '\t' && i && (sample[i - 1] == '\t' || sample[i - 1] == '\n');
    }
    c->avg_row = take / (lines ? lines : 1);
    // more than half of the fields empty: pass 1 gives tabs that directly follow a boundary no descriptor (k_pass1.hip)
    c->skip_empty = 2 * empties > lines + tabs;
    // field boundaries per 2 KiB round of pass 1: tables of mostly empty fields get the large descriptor ring
    c->dense_boundaries = take && (lines + tabs) * 2048 / take > 330;
    return ZDW_OK;
}

// a new input: everything that persists across the blocks of one input starts over
void reset_input_state(zdw_ctx *c)
{
    reset_block(c);
    c->lo = c->next_lo = 0;
    c->row_cap = c->first_row_cap;          // a new input starts with a fresh line buffer (ConvertToZDW.cpp:965)
    c->model_rows_prev = c->model_rows_prev2 = 0;
    c->entries_per_row = 0.0;
    c->row_cap_block_lo = ~0ull;
    c->bytes_per_row = 0.0; c->no_fold = false;
    c->hi = 0;
    c->avg_row = 0;
    c->dense_boundaries = false;
    c->input_open = false;
}

int after_input(zdw_ctx *c)
{
    reset_input_state(c);
    if (c->n) {
        HIP_TRY(c, c->d_scalar.ensure(64));
        launch_find_effective_end(c->d_in, c->n, 0, c->d_scalar.as<uint64_t>(), c->stream);
        HIP_TRY(c, hipMemcpyAsync(&c->hi, c->d_scalar.p, 8, hipMemcpyDeviceToHost, c->stream));
        int rc = sample_input(c);
        if (rc) return rc;
    }
    c->have_input = true;
    return ZDW_OK;
}

// ---- streamed input ----
void stream_fail(InputStream *is, int code, const std::string &text)
{
    std::lock_guard<std::mutex> g(is->m);
    if (!is->error) { is->error = code; is->error_text = text; }
    is->final = true;
    is->cv.notify_all();
}

void stream_feeder(InputStream *is)
{
    (void)hipSetDevice(is->device);
    hipStream_t s = nullptr;
    constexpr int NP = 3;
    void *piece[NP] = {nullptr, nullptr, nullptr};
    hipEvent_t ev[NP] = {nullptr, nullptr, nullptr};
    bool issued[NP] = {false, false, false};
    uint64_t upto[NP] = {0, 0, 0};
    auto hip_ok = [&](hipError_t e, const char *what) {
        if (e == hipSuccess) return true;
        (void)hipGetLastError();
        stream_fail(is, 2, std::string(what) + ": " + hipGetErrorString(e));
        return false;
    };
    bool ok = hip_ok(hipStreamCreateWithFlags(&s, hipStreamNonBlocking), "input stream");
    if (ok && is->host) {
        // host memory of the caller's: copied from where it lies, piece by piece, each published as it lands
        void *img = nullptr;
        ok = hip_ok(hipMalloc(&img, is->size_hint + 64), "device image of the input");
        if (ok) { std::lock_guard<std::mutex> g(is->m); is->img = img; is->cap = is->size_hint + 64; }
        uint64_t off = 0;
        while (ok && off < is->size_hint) {
            { std::lock_guard<std::mutex> g(is->m); if (is->stop) break; }
            const uint64_t n = std::min<uint64_t>(is->piece_bytes, is->size_hint - off);
            ok = hip_ok(hipMemcpyAsync(static_cast<uint8_t *>(img) + off, is->host + off, n, hipMemcpyHostToDevice, s), "input copy") &&
                 hip_ok(hipStreamSynchronize(s), "input copy");
            if (!ok) break;
            off += n;
            std::lock_guard<std::mutex> g(is->m);
            is->uploaded = off;
            is->cv.notify_all();
        }
        if (ok) {
            std::lock_guard<std::mutex> g(is->m);
            is->final = off == is->size_hint;
            if (!is->final && !is->error) { is->error = 2; is->error_text = "stopped"; is->final = true; }
            is->cv.notify_all();
        }
        if (s) (void)hipStreamDestroy(s);
        return;
    }
    for (int i = 0; ok && i < NP; ++i) {
        ok = hip_ok(hipHostMalloc(&piece[i], is->piece_bytes, hipHostMallocDefault), "pinned input piece") &&
             hip_ok(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming), "input event");
    }
    auto publish = [&](uint64_t bytes) {
        std::lock_guard<std::mutex> g(is->m);
        is->uploaded = bytes;
        is->cv.notify_all();
    };
    uint64_t off = 0;
    bool eof = false;
    for (int k = 0; ok && !eof; ++k) {
        const int p = k % NP;
        if (issued[p]) {                                    // the piece is free again once its copy has landed: those bytes are there
            ok = hip_ok(hipEventSynchronize(ev[p]), "input copy");
            if (!ok) break;
            publish(upto[p]);
            issued[p] = false;
        }
        { std::lock_guard<std::mutex> g(is->m); if (is->stop) break; }
        uint64_t got = 0;
        while (got < is->piece_bytes) {
            if (got) { std::lock_guard<std::mutex> g(is->m); if (is->stop) { ok = false; break; } }      // (a slow producer: do not fill the piece first)
            const int64_t r = is->read(is->user, static_cast<uint8_t *>(piece[p]) + got, is->piece_bytes - got);
            if (r < 0) { stream_fail(is, 1, "the input's read callback failed"); ok = false; break; }
            if (r == 0) { eof = true; break; }
            got += (uint64_t)r;
        }
        if (!ok || !got) break;
        if (off + got + 64 > is->cap) {
            // a larger image: everything that has been sent so far moves over (this stream has carried all of it)
            uint64_t want = is->size_hint > off + got ? is->size_hint + 64 : std::max<uint64_t>({2 * is->cap, off + got + 64, is->first_cap});
            void *bigger = nullptr;
            ok = hip_ok(hipMalloc(&bigger, want), "device image of the input");
            if (!ok) break;
            if (off) ok = hip_ok(hipMemcpyAsync(bigger, is->img, off, hipMemcpyDeviceToDevice, s), "input image copy");
            ok = ok && hip_ok(hipStreamSynchronize(s), "input image copy");
            if (!ok) { (void)hipFree(bigger); break; }
            std::lock_guard<std::mutex> g(is->m);
            if (is->img) is->retired.push_back(is->img);
            is->img = bigger;
            is->cap = want;
        }
        ok = hip_ok(hipMemcpyAsync(static_cast<uint8_t *>(is->img) + off, piece[p], got, hipMemcpyHostToDevice, s), "input copy") &&
             hip_ok(hipEventRecord(ev[p], s), "input event");
        if (!ok) break;
        off += got;
        issued[p] = true;
        upto[p] = off;
    }
    if (ok && s) ok = hip_ok(hipStreamSynchronize(s), "input copy");
    if (ok) {
        std::lock_guard<std::mutex> g(is->m);
        is->uploaded = off;
        is->final = true;
        is->cv.notify_all();
    }
    for (int i = 0; i < NP; ++i) { if (piece[i]) (void)hipHostFree(piece[i]); if (ev[i]) (void)hipEventDestroy(ev[i]); }
    if (s) (void)hipStreamDestroy(s);
}

void stream_close(zdw_ctx *c)
{
    InputStream *is = c->in_stream;
    if (!is) return;
    { std::lock_guard<std::mutex> g(is->m); is->stop = true; }
    if (is->th.joinable()) is->th.join();
    for (void *p : is->retired) (void)hipFree(p);
    if (is->img) (void)hipFree(is->img);
    delete is;
    c->in_stream = nullptr;
    c->input_open = false;
    c->d_in = nullptr; c->n = 0; c->have_input = false;
}

// The image as far as it has arrived, at least `need` bytes of it unless the input ends earlier.
int stream_snapshot(zdw_ctx *c, uint64_t need)
{
    InputStream *is = c->in_stream;
    const void *img = nullptr;
    uint64_t len = 0;
    bool final = false;
    {
        std::unique_lock<std::mutex> lk(is->m);
        is->cv.wait(lk, [&] { return is->error || is->final || is->uploaded >= need; });
        if (is->error) return fail(c, ZDW_UNKNOWN_ERROR, "input stream: %s", is->error_text.c_str());
        img = is->img; len = is->uploaded; final = is->final;
    }
    if (len >= EXTRA_BASE) return fail(c, ZDW_OUT_OF_MEMORY, "input of %llu bytes is beyond this build's 2^39 limit", (unsigned long long)len);
    const bool first = c->n == 0 && len != 0;
    c->d_in = static_cast<const uint8_t *>(img);
    c->n = len;
    c->input_open = !final;
    c->hi = c->lo;
    if (len > c->lo) {
        HIP_TRY(c, c->d_scalar.ensure(64));
        launch_find_effective_end(c->d_in, c->n, c->lo, c->d_scalar.as<uint64_t>(), c->stream);
        HIP_TRY(c, hipMemcpyAsync(&c->hi, c->d_scalar.p, 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (first) return sample_input(c);
    return ZDW_OK;
}

// the buffers of the radix sort for u pairs
int sort_buffers(zdw_ctx *c, uint64_t u, SortBuffers &b)
{
    const size_t nb = sort_blocks(u ? u : 1);
    HIP_TRY(c, c->key_a.ensure((u + 1) * 8)); HIP_TRY(c, c->key_b.ensure((u + 1) * 8));
    HIP_TRY(c, c->val_a.ensure((u + 1) * 4)); HIP_TRY(c, c->val_b.ensure((u + 1) * 4));
    HIP_TRY(c, c->order.ensure((u + 1) * 4)); HIP_TRY(c, c->group.ensure((u + 1) * 4));
    HIP_TRY(c, c->active_a.ensure((u + 1) * 4)); HIP_TRY(c, c->active_b.ensure((u + 1) * 4));
    HIP_TRY(c, c->flags.ensure((6 * u + 8) * 4));
    HIP_TRY(c, c->hist.ensure((2 * (256 * nb + 1) + SORT_SPARE_WORDS) * 4));      // (+ dict_sort's spare words)
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(std::max<uint64_t>(256 * nb + 1, std::max<uint64_t>(u + 1, c->nrows + 1ull)))));
    b.key_a = c->key_a.as<uint64_t>(); b.key_b = c->key_b.as<uint64_t>();
    b.val_a = c->val_a.as<uint32_t>(); b.val_b = c->val_b.as<uint32_t>();
    b.order = c->order.as<uint32_t>(); b.group = c->group.as<uint32_t>();
    b.active_a = c->active_a.as<uint32_t>(); b.active_b = c->active_b.as<uint32_t>();
    b.flags = c->flags.as<uint32_t>(); b.hist = c->hist.as<uint32_t>();
    b.scan_tmp = c->scan_tmp.p; b.counter = c->d_scalar.as<uint32_t>();
    return ZDW_OK;
}

// exact string sort of u strings (by id: pos / len, length 0 = hole) into c->order (holes first)
int sort_strings(zdw_ctx *c, const uint8_t *in, uint64_t in_bytes, const uint8_t *extra, const uint64_t *upos, const uint32_t *ulen, uint64_t u,
                 uint32_t max_len, const uint64_t *ukey = nullptr)
{
    hipStream_t s = c->stream;
    StageScope ts(c, ZDW_STAGE_DICT_SORT);
    SortBuffers b{};
    int rcb = sort_buffers(c, u, b);
    if (rcb) return rcb;
    const int e = dict_sort(in, in_bytes, extra, upos, ulen, u, b, s, &c->sort_rounds, max_len, ukey);
    if (e) return fail(c, ZDW_UNKNOWN_ERROR, "dictionary sort: %s", hipGetErrorString((hipError_t)e));
    return ZDW_OK;
}

// ---- dictionary: table -> sorted blob + slot offsets -----------------------------------------
int start_pass2_size(zdw_ctx *c);

// how many unique ids pass 1 handed out, and Dictionary::getSize(): known without sorting anything
int ensure_dict_size(zdw_ctx *c)
{
    if (c->dict_size_done) return ZDW_OK;
    if (!c->p1_done) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_block_pass1 has not run for this block");
    hipStream_t s = c->stream;
    // pass 1 (and the shard import) left the uniques by id: uniq_pos / uniq_len, length 0 = hole
    P1Status st = c->st_host;
    if (!c->st_known) {
        HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    const uint32_t count = st.id_next;
    c->dict_max_len = st.max_len > 24 ? st.max_len : 24;
    if (count > c->id_cap) return fail(c, ZDW_UNKNOWN_ERROR, "unique id counter (%u) beyond its table (%u)", count, c->id_cap);
    c->nuniq = st.entries;
    c->nids = count;
    const uint64_t u = count;
    HIP_TRY(c, c->lenp1.ensure((u + 1) * 4));
    HIP_TRY(c, c->off_excl.ensure((u + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(u + 1)));
    HIP_TRY(c, c->slot_off.ensure((u + 1) * 4));
    // (by id: the sum does not depend on the order.  One launch and the partial sums read back, not lengths + a three-launch scan:
    // this sits between pass 1 and everything that follows it)
    uint64_t total = 0;
    {
        static_assert(DICT_TOTAL_BLOCKS * 8 <= 16384, "the partial sums live at the start of the scan scratch");
        HIP_TRY(c, c->scan_tmp.ensure(std::max<size_t>(scan_tmp_bytes(u + 1), DICT_TOTAL_BLOCKS * 8)));
        const uint32_t nb = launch_dict_total(c->uniq_len.as<uint32_t>(), u, static_cast<uint64_t *>(c->scan_tmp.p), s);
        std::vector<uint64_t> part(nb);
        HIP_TRY(c, hipMemcpyAsync(part.data(), c->scan_tmp.p, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        for (uint64_t v : part) total += v;
    }
    c->dict_size = total + 1;                       // Dictionary::getSize(): origin byte included
    {
        const uint64_t limit = c->max_dict_bytes && c->max_dict_bytes < (1ull << 32) ? c->max_dict_bytes : (1ull << 32) - 1;
        if (c->dict_size > limit)                   // (known before the sort: a block that has to be split costs no sort)
            return fail(c, ZDW_OUT_OF_MEMORY, "dictionary of %llu bytes exceeds the limit of %llu: split the block (smaller zdw_split_hint.max_rows)",
                        (unsigned long long)c->dict_size, (unsigned long long)limit);
    }
    c->index_size = (uint8_t)bytes_for(c->dict_size);
    c->dict_size_done = true;
    return ZDW_OK;
}

int ensure_dict(zdw_ctx *c)
{
    if (c->dict_done) return ZDW_OK;
    int rc = ensure_dict_size(c);
    if (rc) return rc;
    hipStream_t s = c->stream;
    const uint64_t u = c->nids;
    // Pass 2's row lengths need the dictionary's SIZE (the width of text columns), not its order: they are computed on
    // the second stream while the strings are sorted on this one.  (A shard of a row-sharded block waits for its halo row.)
    if (!c->have_prev && !experiment("ZDW_NO_OVERLAP")) { rc = start_pass2_size(c); if (rc) return rc; }
    {
        rc = sort_strings(c, c->d_in, c->n, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), u, c->dict_max_len,
                          c->uniq_key.as<uint64_t>());
        if (rc) return rc;
    }
    {
        StageScope ts(c, ZDW_STAGE_DICT_EMIT);
        launch_dict_lengths(c->order.as<uint32_t>(), c->uniq_len.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);
        launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
        HIP_TRY(c, c->dict_sec.ensure(16 + c->dict_size));
        uint8_t *blob = c->dict_sec.as<uint8_t>() + 8;
        launch_dict_emit(c->d_in, c->n, c->extra.as<uint8_t>(), c->order.as<uint32_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(),
                         c->off_excl.as<uint64_t>(), u, blob, c->slot_off.as<uint32_t>(), s, c->uniq_key.as<uint64_t>());
        if (c->nuniq == 0) {                                   // empty set: a single 00 (dictionary.cpp:80-84)
            c->dict_sec_off = 8; c->dict_sec_len = 1;
        } else {
            uint8_t hdr[8];
            hdr[0] = c->index_size;
            for (unsigned i = 0; i < c->index_size; ++i) hdr[1 + i] = (uint8_t)(c->dict_size >> (8 * i));
            c->dict_sec_off = 8 - 1 - c->index_size;
            c->dict_sec_len = 1 + c->index_size + c->dict_size;
            HIP_TRY(c, hipMemcpyAsync(c->dict_sec.as<uint8_t>() + c->dict_sec_off, hdr, 1 + c->index_size, hipMemcpyHostToDevice, s));
        }
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    c->dict_done = true;
    return ZDW_OK;
}

// ---- where would the reference have ended this block?  (--mem-limit as a model: DESIGN.md) ----
// vsize_MB = 2.6 + 64 x arenas + 64 B x strings / 2^20, checked whenever StringHeap takes a new 64 MiB arena; the strings
// enter the arenas in the order pass 1 first meets them.  Pass 1 has just run over rows [0, nrows): *found = the limit
// trips inside them, at field *fields - 1 of row *row (that row and everything after it belong to the next block).
// The order of first occurrences is taken from uniq_pos: while the model drives pass 1 every occurrence of a string takes part
// in a minimum there (Pass1Args.track_first), so it is the first one whichever occurrence won the insertion.
constexpr double MODEL_BASE_MB = 2.6, MODEL_NODE_BYTES = 64.0;
constexpr uint64_t MODEL_ARENA = 64ull << 20;

int locate_occurrence(zdw_ctx *c, uint64_t pos, uint32_t span_shift, uint64_t span0, uint32_t *row, uint32_t *fields, uint64_t *row_start);

int model_trip(zdw_ctx *c, uint32_t span_shift, uint64_t span0, bool *found, uint32_t *row, uint32_t *fields, uint64_t *row_start, uint64_t *trip_pos)
{
    *found = false;
    hipStream_t s = c->stream;
    P1Status st{};
    HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const uint64_t u = st.id_next, nreal = st.entries;
    if (!nreal || u > c->id_cap) return ZDW_OK;
    // cheap bound first: even with every arena wasting its tail the limit stays out of reach
    HIP_TRY(c, c->lenp1.ensure((u + 1) * 4));
    HIP_TRY(c, c->off_excl.ensure((u + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(u + 1)));
    HIP_TRY(c, c->d_scalar.ensure(64));
    launch_dict_lengths(nullptr, c->uniq_len.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->off_excl.as<uint64_t>() + u, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const uint64_t max_len = std::max<uint64_t>(st.max_len, 24) + 1;
    const double arenas_ub = (double)(total / (MODEL_ARENA > 2 * max_len ? MODEL_ARENA - max_len : MODEL_ARENA / 2) + 1);
    c->model_vsize_end = MODEL_BASE_MB + 64.0 * arenas_ub + MODEL_NODE_BYTES * (double)nreal / 1048576.0;
    if (c->model_vsize_end < c->mem_limit_mb) return ZDW_OK;
    // exact: unique ids by position, prefix sums of len + 1 in that order, the walk over the arenas
    SortBuffers b{};
    int rc = sort_buffers(c, u, b);
    if (rc) return rc;
    uint32_t *by_pos = nullptr;
    launch_model_sort(c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), (uint32_t)u, b, &by_pos, s);
    launch_dict_lengths(by_pos, c->uniq_len.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
    launch_model_trip(c->off_excl.as<uint64_t>(), (uint32_t)nreal, MODEL_BASE_MB, MODEL_NODE_BYTES, c->mem_limit_mb, MODEL_ARENA, c->d_scalar.as<uint64_t>(), s);
    uint64_t res[2] = {~0ull, 0};
    HIP_TRY(c, hipMemcpyAsync(res, c->d_scalar.p, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (res[0] == ~0ull) return ZDW_OK;
    uint32_t id = 0;
    uint64_t pos = 0;
    HIP_TRY(c, hipMemcpy(&id, by_pos + res[0], 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(&pos, c->uniq_pos.as<uint64_t>() + id, 8, hipMemcpyDeviceToHost));
    pos &= POS_MASK;
    if (pos >= EXTRA_BASE) return ZDW_OK;                  // an imported string (multi-GPU merge): no position in this image
    *found = true;
    *trip_pos = pos;
    return locate_occurrence(c, pos, span_shift, span0, row, fields, row_start);
}

// the row (within the block pass 1 has just parsed) and the field that hold input position `pos`
int locate_occurrence(zdw_ctx *c, uint64_t pos, uint32_t span_shift, uint64_t span0, uint32_t *row, uint32_t *fields, uint64_t *row_start)
{
    hipStream_t s = c->stream;
    launch_locate_field(c->d_in, c->lo, pos, c->d_scalar.as<uint64_t>(), s);
    uint64_t loc[2] = {0, 0};
    HIP_TRY(c, hipMemcpyAsync(loc, c->d_scalar.p, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const uint64_t rs = loc[0];
    const uint64_t si = (rs >> span_shift) - span0;        // the span the row starts in; rows that start before it: span_rowbase
    uint32_t rowbase = 0, in_span = 0;
    if (c->last_fold) {
        // pass 1 counted its own rows: a word per span, the rows that start up to and including it (Pass1Args.span_state)
        uint64_t w = 0;
        if (si) HIP_TRY(c, hipMemcpy(&w, c->span_state.as<uint64_t>() + (si - 1), 8, hipMemcpyDeviceToHost));
        rowbase = (uint32_t)(w & 0x3fffffffffffffffull);
    } else {
        HIP_TRY(c, hipMemcpy(&rowbase, c->span_rowbase.as<uint32_t>() + si, 4, hipMemcpyDeviceToHost));
    }
    if (rs > std::max<uint64_t>((span0 + si) << span_shift, c->lo)) {
        // ... and the rows that start inside it before this one: the row scan's own kernel on that piece
        launch_count_row_starts(c->d_in, c->n, c->lo, rs, span0 + si, 1, span_shift, c->d_scalar.as<uint32_t>() + 8, c->status.as<P1Status>(), s);
        HIP_TRY(c, hipMemcpyAsync(&in_span, c->d_scalar.as<uint32_t>() + 8, 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
    }
    *row = rowbase + in_span;
    *fields = (uint32_t)loc[1] + 1;
    *row_start = rs;
    return ZDW_OK;
}

int ensure_dense_rec(zdw_ctx *c);

// The block that pass 1 has just parsed ends earlier, at field `fields` of row `row` (which starts at input position row_start and
// has a record: row < the rows pass 1 kept).  Instead of a second pass 1 up to there, what the rows beyond that point added is
// taken back (k_trim_scan: launch_trim_block): strings only they hold become holes among the unique ids, ranges and used columns
// are rebuilt from the records that stay, the line buffer is replayed up to the new end.
// trip_pos: where the field that ends the block starts, when the caller knows it (~0: not).  With every string's first position at
// hand (the model's pass 1 keeps them) the strings that stay are those first met at or before it, and the scan of the records
// need not mark them one by one (58 M single-byte stores all over a 40 MB array per block of C5).
int trim_block(zdw_ctx *c, uint32_t row, uint32_t fields, uint64_t row_start, zdw_pass1_result *res, bool allow_no_rows = false, uint64_t trip_pos = ~0ull)
{
    hipStream_t s = c->stream;
    const bool sparse = c->nz_mode && !c->rec_dense;      // (the matrix means something only where the map says so: the scan goes by the map too)
    const bool by_pos = trip_pos != ~0ull && c->last_track && !test_hook("ZDW_TRIM_BY_SCAN");      // (tests: the other form)
    P1Status st{};
    HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const uint32_t u = st.id_next;
    if (!by_pos) {
        HIP_TRY(c, c->lenp1.ensure((uint64_t)u + 64));               // a byte per unique id: some kept row refers to it
        HIP_TRY(c, hipMemsetAsync(c->lenp1.p, 0, (uint64_t)u + 64, s));
    }
    HIP_TRY(c, hipMemsetAsync(c->col_min.p, 0xFF, c->ncols * 8 + 8, s));
    HIP_TRY(c, hipMemsetAsync(c->col_max.p, 0, c->ncols * 8 + 8, s));
    HIP_TRY(c, hipMemsetAsync(c->col_seen.p, 0, c->ncols + 8, s));
    uint32_t *d_entries = &c->status.as<P1Status>()->entries;
    c->st_known = false;
    HIP_TRY(c, hipMemsetAsync(d_entries, 0, 4, s));
    {
        StageScope ts(c, ZDW_STAGE_PASS1);
        launch_trim_block(c->rec.as<uint8_t>(), c->rec_stride, c->ncols, c->d_kind.as<uint8_t>(), c->d_recoff.as<uint32_t>(), row, fields,
                          by_pos ? nullptr : c->lenp1.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), trip_pos,
                          c->col_min.as<uint64_t>(), c->col_max.as<uint64_t>(), c->col_seen.as<uint8_t>(),
                          c->uniq_len.as<uint32_t>(), u, d_entries, sparse ? c->nz.as<uint32_t>() : nullptr, c->nz_stride, s);
    }
    HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->st_host = st; c->st_known = true;
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail(c, ZDW_UNKNOWN_ERROR, "block trim launch: %s", hipGetErrorString(le));
    c->nrows = row;
    c->is_last = false;
    c->partial_taken = fields;
    c->next_lo = row_start;
    c->row_cap = c->row_cap_block;                                    // the line buffer: again from the block's start, up to the new end
    c->long_read_end = row_start;
    c->long_tail = false;
    int rc = replay_long_rows(c, &c->row_cap);
    if (rc) return rc;
    res->num_rows = row;
    res->is_last_block = 0;
    res->longest_line = c->row_cap;
    res->bytes_consumed = row_start;
    res->dict_entries = st.entries;
    if (c->ncols) {
        HIP_TRY(c, hipMemcpy(c->h_min.data(), c->col_min.p, c->ncols * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_max.data(), c->col_max.p, c->ncols * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_seen.data(), c->col_seen.p, c->ncols, hipMemcpyDeviceToHost));
    }
    if (row == 0 && !allow_no_rows) return fail(c, ZDW_OUT_OF_MEMORY, "block limit leaves no room for a single row");
    return ZDW_OK;
}

// the block's ranges, dictionary or halo row have changed: row lengths computed ahead (start_pass2_size) are void
// keep_early: the caller changes nothing the early row lengths of a shard depend on (zdw_shard_rows_early)
void invalidate_pass2(zdw_ctx *c, bool keep_early = false)
{
    if (c->p2_size_done && !(keep_early && c->p2_size_early)) {
        (void)hipStreamSynchronize(c->stream2);
        c->p2_size_done = c->p2_size_early = c->p2_row0_stale = false;
    }
    c->p2_done = false;
}

// ---- column widths and bases (writeLookupColumnStats, ConvertToZDW.cpp:417-483) ----------------
int ensure_stats(zdw_ctx *c)
{
    if (c->stats_done) return ZDW_OK;
    int rc = c->dict_done ? ZDW_OK : ensure_dict_size(c);
    if (rc) return rc;
    c->col_size.assign(c->ncols, 0);
    c->col_base.assign(c->ncols, 0);
    c->used.clear();
    for (uint32_t k = 0; k < c->ncols; ++k) {
        if (c->kind[k] == K_TEXT) {
            if (!c->h_seen[k]) continue;
            c->col_size[k] = c->index_size; c->col_base[k] = 0;
        } else {
            if (c->h_max[k] == 0) continue;                       // no value > 0 seen: column unused
            const uint64_t base = c->h_min[k] - 1;
            c->col_base[k] = base;
            c->col_size[k] = (uint8_t)bytes_for(c->h_max[k] - base);
        }
        c->used.push_back(k);
    }
    c->stats_done = true;
    return ZDW_OK;
}

// uploads the per-used-column tables pass 2 reads and fills the kernel arguments
// A reader that wants every record of the matrix (the column kernels of pass 2, the halo row of a shard, a block cut back to the
// model's end): what pass 1 left unwritten under Pass1Args.nz is zeroed now.
int ensure_dense_rec(zdw_ctx *c)
{
    if (!c->nz_mode || c->rec_dense) return ZDW_OK;
    launch_rec_zero_fill(c->rec.as<uint8_t>(), c->rec_stride, c->nrows_written, c->ncols, c->d_kind.as<uint8_t>(), c->d_recoff.as<uint32_t>(),
                         c->nz.as<uint32_t>(), c->nz_stride, c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->rec_dense = true;
    return ZDW_OK;
}

int prepare_pass2(zdw_ctx *c, Pass2Args &a, hipStream_t s = nullptr)
{
    int rc = ensure_stats(c);
    if (rc) return rc;
    if (!s) s = c->stream;
    const uint32_t nu = (uint32_t)c->used.size();
    memset(&a, 0, sizeof a);
    a.nrows = c->nrows; a.nused = nu; a.rec_stride = c->rec_stride;
    if (nu == 0 || c->nrows == 0) return ZDW_OK;
    std::vector<uint32_t> uo(nu); std::vector<uint8_t> uk(nu), us(nu); std::vector<uint64_t> ub(nu), up(nu, 0);
    for (uint32_t u = 0; u < nu; ++u) {
        const uint32_t k = c->used[u];
        uo[u] = c->recoff[k]; uk[u] = c->kind[k]; us[u] = c->col_size[k]; ub[u] = c->col_base[k];
        if (c->have_prev) up[u] = c->prev_row[k];
        a.out_row_max += us[u];
    }
    a.out_row_max += (nu + 7) / 8;
    HIP_TRY(c, c->d_used_recoff.ensure(nu * 4)); HIP_TRY(c, c->d_used_kind.ensure(nu)); HIP_TRY(c, c->d_used_size.ensure(nu));
    HIP_TRY(c, c->d_used_base.ensure(nu * 8)); HIP_TRY(c, c->d_prev0.ensure(nu * 8)); HIP_TRY(c, c->d_last.ensure(nu * 8));
    HIP_TRY(c, hipMemcpyAsync(c->d_used_recoff.p, uo.data(), nu * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_used_kind.p, uk.data(), nu, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_used_size.p, us.data(), nu, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_used_base.p, ub.data(), nu * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_prev0.p, up.data(), nu * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));      // the host vectors above go out of scope
    a.rec = c->rec.as<uint8_t>();
    a.used_recoff = c->d_used_recoff.as<uint32_t>(); a.used_kind = c->d_used_kind.as<uint8_t>();
    a.used_size = c->d_used_size.as<uint8_t>(); a.used_base = c->d_used_base.as<uint64_t>();
    a.id_off = c->slot_off.as<uint32_t>(); a.nids = c->nids; a.prev0 = c->d_prev0.as<uint64_t>();
    a.prev_is_zero = c->have_prev ? 0u : 1u;
    a.last_stored = c->d_last.as<uint64_t>();
    if (c->nz_mode && !c->rec_dense) {
        // rows by the map of non-zero records (k_pass2_rows_sparse) when a row's candidates fit its LDS; else the whole matrix
        const size_t per_wave = (((size_t)nu * 2 + 15) & ~(size_t)15) + ((((size_t)nu + 7) / 8 + 15) & ~(size_t)15) + 16 + 1024;      // (k_pass2.hip: p2s_*_bytes)
        if (per_wave * 4 <= 60 * 1024 && c->ncols <= 16384 && !test_hook("ZDW_P2_DENSE")) {
            // per schema column: which used column it is and that column's facts; then per used column its schema column
            std::vector<uint64_t> maps((size_t)c->ncols * 2 + (nu + 1) / 2, 0xFFFFFFull);
            uint32_t *ucol = reinterpret_cast<uint32_t *>(maps.data() + (size_t)c->ncols * 2);
            for (uint32_t u = 0; u < nu; ++u) {
                maps[c->used[u]] = (uint64_t)u | (uint64_t)(uk[u] & 3u) << 24 | (uint64_t)(us[u] & 15u) << 26 | (uint64_t)uo[u] << 32;
                maps[(size_t)c->ncols + c->used[u]] = ub[u];
                ucol[u] = c->used[u];
            }
            HIP_TRY(c, c->d_used_col.ensure(maps.size() * 8));
            HIP_TRY(c, hipMemcpyAsync(c->d_used_col.p, maps.data(), maps.size() * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipStreamSynchronize(s));
            a.nz = c->nz.as<uint32_t>(); a.nz_stride = c->nz_stride;
            a.col_meta = c->d_used_col.as<uint64_t>(); a.nz_cols = c->ncols; a.used_col = reinterpret_cast<const uint32_t *>(a.col_meta + (size_t)c->ncols * 2);
        } else {
            rc = ensure_dense_rec(c);
            if (rc) return rc;
        }
    }
    return ZDW_OK;
}

int pass2_size_buffers(zdw_ctx *c, Pass2Args &a)
{
    HIP_TRY(c, c->row_len.ensure((c->nrows + 1ull) * 4));
    HIP_TRY(c, c->row_off.ensure((c->nrows + 2ull) * 8));
    HIP_TRY(c, c->scan_tmp2.ensure(scan_tmp_bytes(c->nrows + 1ull)));
    HIP_TRY(c, c->batch_pre.ensure(((uint64_t)c->nrows + 1) * ((a.nused + 63) / 64) * 2));
    a.row_len = c->row_len.as<uint32_t>(); a.row_off = c->row_off.as<uint64_t>(); a.out = nullptr;
    a.batch_pre = c->batch_pre.as<uint16_t>();
    return ZDW_OK;
}

// row lengths + their scan on the second stream (no dictionary offset is needed for them: ensure_dict)
int start_pass2_size(zdw_ctx *c)
{
    if (c->p2_size_done || c->p2_done) return ZDW_OK;
    Pass2Args a;
    int rc = prepare_pass2(c, a, c->stream2);
    if (rc) return rc;
    if (a.nused == 0 || c->nrows == 0) return ZDW_OK;
    rc = pass2_size_buffers(c, a);
    if (rc) return rc;
    {
        StageScope ts(c, ZDW_STAGE_PASS2_SIZE, c->stream2);
        // (a smaller grid beside the sort -- unless the record matrix is so large that the row lengths, not the sort, are the longer
        // of the two: wide sparse tables)
        a.background = (uint64_t)c->nrows * c->rec_stride <= (8ull << 30) ? 1u : 0u;
        launch_pass2_size(a, c->stream2);
        launch_scan_u32_to_u64(a.row_len, c->row_off.as<uint64_t>(), c->nrows, c->scan_tmp2.p, nullptr, c->stream2);
    }
    c->p2_size_done = true;
    return ZDW_OK;
}

int run_pass2(zdw_ctx *c)
{
    if (c->p2_done) return ZDW_OK;
    int rc = ensure_dict(c);                    // the rows store dictionary offsets
    if (rc) return rc;
    if (c->p2_size_done) HIP_TRY(c, hipStreamSynchronize(c->stream2));      // row lengths started while the dictionary was being sorted
    Pass2Args a;
    rc = prepare_pass2(c, a);
    if (rc) return rc;
    hipStream_t s = c->stream;
    c->rows_len = 0;
    if (a.nused == 0 || c->nrows == 0) { c->p2_done = true; HIP_TRY(c, c->rows_out.ensure(16)); return ZDW_OK; }
    rc = pass2_size_buffers(c, a);
    if (rc) return rc;
    if (!c->p2_size_done) {
        StageScope ts(c, ZDW_STAGE_PASS2_SIZE);
        launch_pass2_size(a, s);
        launch_scan_u32_to_u64(a.row_len, c->row_off.as<uint64_t>(), c->nrows, c->scan_tmp2.p, nullptr, s);
    } else if (c->p2_row0_stale) {
        // the row lengths were computed early, as if nothing preceded the shard's first row: that row again, against the row before it,
        // and the offsets once more
        StageScope ts(c, ZDW_STAGE_PASS2_SIZE);
        Pass2Args b = a;
        b.nrows = 1; b.background = 0;
        launch_pass2_size(b, s);
        launch_scan_u32_to_u64(a.row_len, c->row_off.as<uint64_t>(), c->nrows, c->scan_tmp2.p, nullptr, s);
        c->p2_row0_stale = false;
    }
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->row_off.as<uint64_t>() + c->nrows, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, c->rows_out.ensure(total + 16));
    a.out = c->rows_out.as<uint8_t>();
    {
        StageScope ts(c, ZDW_STAGE_PASS2_EMIT);
        launch_pass2_emit(a, s);
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    c->rows_len = total;
    c->p2_done = true;
    return ZDW_OK;
}

std::vector<DevBuf *> all_device_buffers(zdw_ctx *c)
{
    return {&c->d_kind, &c->d_recoff, &c->d_numidx, &c->in_owned, &c->span_starts, &c->span_rowbase, &c->span_state, &c->rowidx, &c->rx_bsmap, &c->nz, &c->d_used_col, &c->scan_tmp, &c->scan_tmp2, &c->rec,
                      &c->slots, &c->deferred_q, &c->my_order, &c->part_idx, &c->part_keys, &c->part_bounds, &c->part_split, &c->part_lens, &c->own_slots,
                      &c->own_upos, &c->own_ulen, &c->own_key, &c->own_status, &c->own_extra, &c->own_lens, &c->own_offs, &c->own_ids, &c->own_idoff, &c->own_blob,
                      &c->own_out, &c->own_pos, &c->own_minpos, &c->own_flags, &c->model_len, &c->exp_pos, &c->col_min, &c->col_max, &c->col_seen, &c->status, &c->long_rows, &c->d_scalar,
                      &c->uniq_pos, &c->uniq_len, &c->uniq_key, &c->key_a, &c->key_b, &c->val_a, &c->val_b, &c->order, &c->group,
                      &c->active_a, &c->active_b, &c->flags, &c->hist, &c->lenp1, &c->off_excl, &c->slot_off, &c->dict_sec, &c->extra,
                      &c->extra_lens, &c->extra_offs, &c->exp_bytes, &c->d_used_recoff, &c->d_used_kind, &c->d_used_size,
                      &c->d_used_base, &c->d_prev0, &c->d_last, &c->row_len, &c->row_off, &c->rows_out, &c->batch_pre};
}

}  // namespace

// ================================================================================================
extern "C" {

uint32_t zdw_abi_version(void) { return ZDW_ABI_VERSION; }

const char *zdw_last_error(const zdw_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int zdw_ctx_create(const uint8_t *col_types, uint32_t ncols, const zdw_opts *opts, zdw_ctx **out)
{
    if (!out) return fail(nullptr, ZDW_UNKNOWN_ERROR, "zdw_ctx_create: out is NULL");
    *out = nullptr;
    if (ncols && !col_types) return fail(nullptr, ZDW_UNKNOWN_ERROR, "zdw_ctx_create: col_types is NULL");
    if (ncols >= COL_SKIP) return fail(nullptr, ZDW_UNKNOWN_ERROR, "zdw_ctx_create: too many columns (%u)", ncols);
    zdw_ctx *c = new (std::nothrow) zdw_ctx();
    if (!c) return fail(nullptr, ZDW_OUT_OF_MEMORY, "zdw_ctx_create: out of host memory");
    int dev = opts ? opts->device : -1;
    hipError_t e = hipSuccess;
    if (dev < 0) e = hipGetDevice(&dev); else e = hipSetDevice(dev);
    if (e != hipSuccess) {
        int rc = fail(nullptr, ZDW_UNKNOWN_ERROR, "no usable HIP device: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    c->device = dev;
    {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest);
    }
    if (e != hipSuccess) {
        int rc = fail(nullptr, ZDW_UNKNOWN_ERROR, "hipStreamCreate: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    {   // the second stream yields to the first: what runs on it fills the gaps of the launch-bound dictionary sort
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, least);
    }
    if (e != hipSuccess) {
        int rc = fail(nullptr, ZDW_UNKNOWN_ERROR, "hipStreamCreate: %s", hipGetErrorString(e));
        (void)hipStreamDestroy(c->stream);
        delete c;
        return rc;
    }
    c->ncols = ncols;
    c->types.assign(col_types, col_types + ncols);
    c->kind.resize(ncols); c->recoff.resize(ncols); c->numidx.resize(ncols);
    // rec row layout: u64 slots (numeric columns) first, then u32 slots (text columns); 8-byte aligned stride
    uint32_t off = 0, nnum = 0;
    for (uint32_t k = 0; k < ncols; ++k) {
        c->kind[k] = kind_of_type(col_types[k]);
        if (c->kind[k] != K_TEXT) { c->recoff[k] = off; off += 8; c->numidx[k] = (int32_t)nnum++; }
    }
    for (uint32_t k = 0; k < ncols; ++k)
        if (c->kind[k] == K_TEXT) { c->recoff[k] = off; off += 4; c->numidx[k] = -1; }
    c->rec_stride = (off + 7u) & ~7u;
    if (!c->rec_stride) c->rec_stride = 8;
    c->nnum = nnum;
    if (opts) {
        c->trim = opts->trim_trailing_spaces != 0;
        c->first_row_cap = c->row_cap = opts->first_row_capacity ? opts->first_row_capacity : FIRST_ROW_CAP;
        c->slots_opt = opts->dict_slots;
        c->max_dict_bytes = opts->max_dict_bytes;
        if (opts->struct_size >= offsetof(zdw_opts, max_device_bytes) + sizeof(uint64_t)) c->budget.cap = opts->max_device_bytes;
        if (opts->struct_size >= offsetof(zdw_opts, mem_limit_mb) + sizeof(uint64_t)) c->mem_limit_mb = (double)opts->mem_limit_mb;
        if (opts->struct_size >= offsetof(zdw_opts, track_first_occurrence) + sizeof(uint64_t)) c->track_first_opt = opts->track_first_occurrence != 0;
        if (opts->struct_size >= offsetof(zdw_opts, defer_bad_row) + sizeof(uint64_t)) c->defer_bad_row = opts->defer_bad_row != 0;
    }
    for (DevBuf *b : all_device_buffers(c)) b->bud = &c->budget;
    int rc = upload_schema(c);
    if (rc) { g_create_error = c->err; zdw_ctx_destroy(c); return rc; }
    *out = c;
    return ZDW_OK;
}

void zdw_ctx_destroy(zdw_ctx *c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);
    collect_times(c);
    stream_close(c);
    for (DevBuf *b : all_device_buffers(c)) b->release();
    c->h_dict.release(); c->h_rows.release();
    for (int i = 0; i < 2; ++i) { c->h_piece[i].release(); if (c->piece_ev[i]) (void)hipEventDestroy(c->piece_ev[i]); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
}

int zdw_ctx_enable_timing(zdw_ctx *c, int enable)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    collect_times(c);
    c->timing = enable != 0;
    memset(&c->times, 0, sizeof c->times);
    return ZDW_OK;
}

int zdw_ctx_stage_times(const zdw_ctx *cc, zdw_stage_times *out)
{
    if (!cc || !out) return ZDW_UNKNOWN_ERROR;
    zdw_ctx *c = const_cast<zdw_ctx *>(cc);
    collect_times(c);
    *out = c->times;
    return ZDW_OK;
}

int zdw_set_input_host(zdw_ctx *c, const uint8_t *tsv, uint64_t len)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    if (len && !tsv) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_set_input_host: tsv is NULL");
    if (len >= EXTRA_BASE) return fail(c, ZDW_OUT_OF_MEMORY, "input of %llu bytes is beyond this build's 2^39 limit", (unsigned long long)len);
    (void)hipSetDevice(c->device);
    stream_close(c);
    HIP_TRY(c, c->in_owned.ensure(len + 64));
    {
        StageScope ts(c, ZDW_STAGE_H2D);
        if (len) HIP_TRY(c, hipMemcpyAsync(c->in_owned.p, tsv, len, hipMemcpyHostToDevice, c->stream));
    }
    c->d_in = c->in_owned.as<uint8_t>();
    c->n = len;
    return after_input(c);
}

int zdw_set_input_device(zdw_ctx *c, const void *d_tsv, uint64_t len)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    if (len && !d_tsv) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_set_input_device: d_tsv is NULL");
    if (reinterpret_cast<uintptr_t>(d_tsv) & 15u) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_set_input_device: buffer must be 16-byte aligned");
    if (len >= EXTRA_BASE) return fail(c, ZDW_OUT_OF_MEMORY, "input of %llu bytes is beyond this build's 2^39 limit", (unsigned long long)len);
    (void)hipSetDevice(c->device);
    stream_close(c);
    c->d_in = static_cast<const uint8_t *>(d_tsv);
    c->n = len;
    return after_input(c);
}

int zdw_set_input_stream(zdw_ctx *c, zdw_read_fn read, void *user, uint64_t size_hint)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    if (!read) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_set_input_stream: read is NULL");
    (void)hipSetDevice(c->device);
    stream_close(c);
    c->in_owned.release();
    reset_input_state(c);
    InputStream *is = new (std::nothrow) InputStream;
    if (!is) return fail(c, ZDW_OUT_OF_MEMORY, "out of host memory");
    is->read = read; is->user = user; is->device = c->device; is->size_hint = size_hint;
    if (const char *e = test_hook("ZDW_STREAM_PIECE")) is->piece_bytes = std::max<uint64_t>(4096, strtoull(e, nullptr, 10));      // tests
    if (const char *e = test_hook("ZDW_STREAM_CAP0")) is->first_cap = std::max<uint64_t>(4096, strtoull(e, nullptr, 10));         // tests
    c->in_stream = is;
    c->d_in = nullptr; c->n = 0;
    c->input_open = true;
    c->have_input = true;
    is->th = std::thread(stream_feeder, is);
    return ZDW_OK;
}

int zdw_set_input_host_async(zdw_ctx *c, const uint8_t *tsv, uint64_t len)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    if (len && !tsv) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_set_input_host_async: tsv is NULL");
    if (len >= EXTRA_BASE) return fail(c, ZDW_OUT_OF_MEMORY, "input of %llu bytes is beyond this build's 2^39 limit", (unsigned long long)len);
    (void)hipSetDevice(c->device);
    stream_close(c);
    c->in_owned.release();
    reset_input_state(c);
    InputStream *is = new (std::nothrow) InputStream;
    if (!is) return fail(c, ZDW_OUT_OF_MEMORY, "out of host memory");
    is->host = tsv ? tsv : reinterpret_cast<const uint8_t *>("");
    is->device = c->device; is->size_hint = len;
    is->piece_bytes = std::min<uint64_t>(std::max<uint64_t>(len / 8, 64ull << 20), 1ull << 30);      // (a copy from pageable memory costs ~1 ms to start)
    if (const char *e = test_hook("ZDW_STREAM_PIECE")) is->piece_bytes = std::max<uint64_t>(4096, strtoull(e, nullptr, 10));      // tests
    c->in_stream = is;
    c->d_in = nullptr; c->n = 0;
    c->input_open = true;
    c->have_input = true;
    is->th = std::thread(stream_feeder, is);
    return ZDW_OK;
}

static int pass1_impl(zdw_ctx *c, const zdw_split_hint *hint, zdw_pass1_result *res);

uint64_t zdw_ctx_pass1_launches(const zdw_ctx *c) { return c ? c->pass1_launches : 0; }
uint32_t zdw_block_partial_fields(const zdw_ctx *c) { return c ? c->partial_taken : 0; }

int zdw_block_pass1(zdw_ctx *c, const zdw_split_hint *hint, zdw_pass1_result *res)
{
    if (!c || !res) return ZDW_UNKNOWN_ERROR;
    if (c->mem_limit_mb <= 0.0 || (hint && hint->max_rows != 0xFFFFFFFFu)) return pass1_impl(c, hint, res);      // (a caller's split overrides the model)
    // The reference's block limit as a model (model_trip).  The block is parsed once over a window of rows; when the limit trips
    // inside the window, what the rows beyond that point added is taken back out of pass 1's results (trim_block): one pass-1
    // launch per block, another one only when the window turns out too short.
    // The window, in rows.  After a block the model ended: the rows of that block plus a quarter -- plus a tenth when the two blocks
    // before were of one size (tables of id-like columns end their blocks every N rows), plus a thirty-second when they differed by
    // less than one row in 512 (C5: by one in 10 000).  For an input's first block: the rows
    // that four times the limit's bytes of input hold (C5: a row's id fields cost 0.44 bytes of the modelled process per input
    // byte); when the limit does not trip inside a window, the next one is sized from the process size the model saw at its end.
    if (c->in_stream && c->n == 0) {         // a streamed input's first bytes: the row length the first window is sized by
        (void)hipSetDevice(c->device);
        const int rc0 = stream_snapshot(c, 1);
        if (rc0) return rc0;
    }
    uint64_t window;
    if (c->model_rows_prev) {
        const uint64_t a = c->model_rows_prev, b = c->model_rows_prev2;
        const uint64_t diff = a > b ? a - b : b - a;
        const bool steady = b && diff <= a / 32, very_steady = b && diff <= a / 512;
        window = a + (very_steady ? a / 32 : (steady ? a / 10 : a / 4)) + 1000;
    } else {
        const double bytes = 4.0 * c->mem_limit_mb * 1048576.0;
        window = (uint64_t)(bytes / (double)(c->avg_row ? c->avg_row : 1)) + 1000;
    }
    {   // The limit is only looked at when an arena is taken: a block holds at most the strings that bring the process to the limit
        // ((limit - first arena) / 64 B) plus one more arena of them (64 MiB of strings of a few bytes each: 2^24 as a bound for
        // tables, not for adversaries -- beyond it the block ends where the table is full).  The table never needs more than twice that.
        const double max_nodes = std::max(0.0, c->mem_limit_mb - MODEL_BASE_MB - 64.0) * 1048576.0 / MODEL_NODE_BYTES + (double)(1u << 24);
        uint64_t ms = 1ull << 14;
        while ((double)ms < 2.0 * max_nodes && ms < (1ull << 31)) ms <<= 1;
        c->model_max_slots = ms;
    }
    struct Unset { zdw_ctx *c; ~Unset() { c->model_max_slots = 0; } } unset{c};
    const bool dbg = experiment("ZDW_DEBUG_RETRY") != nullptr;
    for (;;) {
        zdw_split_hint w = {window >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)window, 0};
        const double t0 = now_ms();
        int rc = pass1_impl(c, w.max_rows == 0xFFFFFFFFu ? nullptr : &w, res);
        if (rc) return rc;
        const double t1 = now_ms();
        bool found = false;
        uint32_t row = 0, fields = 0;
        uint64_t row_start = 0, trip_pos = ~0ull;
        c->model_vsize_end = 0.0;
        rc = model_trip(c, c->last_span_shift, c->last_span0, &found, &row, &fields, &row_start, &trip_pos);
        if (rc) return rc;
        if (dbg) fprintf(stderr, "[model] window %llu rows: pass 1 %.1f ms, model %.1f ms (alloc so far %.1f)\n", (unsigned long long)window, t1 - t0, now_ms() - t1, g_alloc_ms);
        if (found && row <= res->num_rows) {
            c->model_rows_prev2 = c->model_rows_prev;
            c->model_rows_prev = row;
            // the rows beyond the end are taken back out of what pass 1 left (trim_block); ZDW_MODEL_RERUN=1 (tests) or an end
            // on the window's last row, which has no record: a second pass 1 up to exactly there instead
            if (row < res->num_rows && !test_hook("ZDW_MODEL_RERUN")) return trim_block(c, row, fields, row_start, res, false, trip_pos);
            zdw_split_hint e = {row, fields};
            return pass1_impl(c, &e, res);
        }
        if (res->is_last_block || res->num_rows < w.max_rows) return ZDW_OK;      // all of the input, or a block that device memory ended
        // the block is longer than the window: again, as wide as the growth seen so far says (with a quarter to spare), at least twice
        double grow = 2.0;
        if (c->model_vsize_end > MODEL_BASE_MB + 64.0)
            grow = std::max(2.0, 1.25 * (c->mem_limit_mb - MODEL_BASE_MB) / (c->model_vsize_end - MODEL_BASE_MB));
        window = (uint64_t)std::min(4.0e9, (double)window * std::min(grow, 64.0)) + 1000;
    }
}

// One pass 1 over the current block, in four steps (a job per call; what the steps share are its members):
//   window_and_rows   which bytes of the image the block may take, and how many rows start in them (the row scan -- or, for a block
//                     that takes all that is left, an estimate: pass 1 then counts its own rows)
//   buffers           the block's row limit; record matrix, first dictionary table, list of long rows
//   parse             the kernels, run again with a larger table / queue / fewer rows until the block fits
//   results           what the block is: rows, end, line buffer, column ranges
// P1_COUNT_FIRST: a pass that counted its own rows cannot go on (more rows than estimated, no room for its matrix): the job is run
// again with a row scan in front (c->no_fold is set).
constexpr int P1_COUNT_FIRST = -1;

struct Pass1Job {
    zdw_ctx *c;
    const zdw_split_hint *hint;
    zdw_pass1_result *res;
    hipStream_t s = nullptr;
    Lap lap;
    uint64_t lo = 0, hi = 0, total_rows = 0, span0 = 0, fold_rows_cap = 0, nslots = 0;
    uint32_t nspans = 0, span_shift = SPAN_SHIFT_MIN, row_limit = 0xFFFFFFFFu, partial = 0, nrows = 0, long_cap = 0;
    bool fold = false, tripped = false, lds_ranges = false;
    P1Status st{};

    int window_and_rows();
    int buffers();
    int parse();
    int results();
    bool fewer_rows();
    uint64_t first_table(uint32_t rows_in_block);
};

// Buffers that grow with the block's rows (records, dictionary table): when the device (or zdw_opts.max_device_bytes)
// cannot hold them the block ends earlier -- half the rows, again if need be -- instead of failing; the reference ends
// a block for the same reason, memory, at a point that depends on the machine (Q9).
bool Pass1Job::fewer_rows()
{
    if (fold) return false;      // (a matrix sized from an estimate did not fit: count the rows first after all)
    if (nrows < 2) return false;
    nrows /= 2; tripped = true; row_limit = nrows; partial = 0;
    return true;
}

// dictionary table (32-byte slots): one slot per 32 input bytes, within [2^14, 2^25] (1 GiB); load factor
// limit 5/8; a context remembers what its last block needed
uint64_t Pass1Job::first_table(uint32_t rows_in_block)
{
    uint64_t ns = c->slots_opt;
    if (const char *e = test_hook("ZDW_DICT_SLOTS")) ns = strtoull(e, nullptr, 10);   // experiments
    if (!ns) {
        ns = 1ull << 14;
        // (the block's share of the remaining input, by rows)
        const uint64_t bytes = total_rows ? (uint64_t)((double)(hi > lo ? hi - lo : 0) * rows_in_block / (double)total_rows) : 0;
        // (a window of the memory-limit model that holds 2 GB of slots' worth of input: a table of 2^26 -- blocks of id-like
        // columns reach the limit at tens of millions of strings, and a table that overflows is a second pass 1 over the window;
        // the larger table costs the benchmark's table 0.1 ms: tools/ab_slots.sh)
        const uint64_t cap = (c->model_max_slots && bytes / 32 >= (1ull << 26)) ? (1ull << 26) : (1ull << 25);
        while (ns < bytes / 32 && ns < cap) ns <<= 1;
        if (c->slots_learned > ns && rows_in_block >= c->rows_learned) ns = c->slots_learned;
        // what the rows of this input have held so far, per row (load limit 5/8, 15 % to spare)
        const double est = (double)rows_in_block * c->entries_per_row * 1.6 * 1.15;
        while ((double)ns < est && ns < (1ull << 31)) ns <<= 1;
        if (c->model_max_slots && ns > c->model_max_slots) ns = c->model_max_slots;
    }
    uint64_t p2 = 1;
    while (p2 < ns) p2 <<= 1;
    return p2;
}

int Pass1Job::window_and_rows()
{
    memset(res, 0, sizeof *res);
    if (!c->have_input) return fail(c, ZDW_UNKNOWN_ERROR, "no input set");
    (void)hipSetDevice(c->device);
    s = c->stream;
    lap.mark("start");
    reset_block(c);
    if (c->oom_seen) {
        // The caller comes back with fewer rows after an allocation failed: buffers sized for the larger attempt would
        // keep the memory the smaller one needs.  Everything but the input image and the schema is rebuilt on demand.
        HIP_TRY(c, hipStreamSynchronize(s));
        for (DevBuf *b : all_device_buffers(c))
            if (b != &c->in_owned && b != &c->d_kind && b != &c->d_recoff && b != &c->d_numidx) b->release();
        c->oom_seen = false;
    }
    lo = c->lo;
    hi = 0;
    // a block that is run again (smaller split hint after an oversized dictionary) starts from the line buffer the block
    // began with: rows that move to the next block must not have grown it
    if (c->row_cap_block_lo == lo) c->row_cap = c->row_cap_block;
    else { c->row_cap_block = c->row_cap; c->row_cap_block_lo = lo; }
    res->longest_line = c->row_cap;
    res->is_last_block = 1;
    res->bytes_consumed = lo;
    c->is_last = true;
    HIP_TRY(c, c->status.ensure(sizeof(P1Status)));
    HIP_TRY(c, c->d_scalar.ensure(64));
    if (c->in_stream && c->n == 0) { int rc0 = stream_snapshot(c, 1); if (rc0) return rc0; }      // (the row length below comes from the first bytes)

    total_rows = 0;
    nspans = 0; span_shift = SPAN_SHIFT_MIN;
    span0 = 0;
    // A caller that wants max_rows rows does not need the rows of everything that is left counted (a file of many blocks: the
    // sum over its blocks of what is left): the image is looked at as far as those rows should reach, by the bytes per row seen so
    // far and a quarter more, and twice as far again while that turns out too short.
    const bool windowed = hint && hint->max_rows != 0xFFFFFFFFu && c->avg_row;
    // (a block before this one says what a row takes: a fiftieth more instead of a quarter)
    const double reach = !windowed ? 0.0 : (c->bytes_per_row > 0.0 ? (double)hint->max_rows * c->bytes_per_row * 1.02 + 1048576.0
                                                                    : (double)hint->max_rows * (double)c->avg_row * 1.25 + 1048576.0);
    // A streamed input: the block is looked at once the bytes it should need have arrived -- everything for a block without a
    // row limit -- and again, with more of them, while the input is still open and the block reaches to the end of what is there.
    uint64_t need = windowed && reach < 1.0e18 ? lo + (uint64_t)reach : ~0ull;
    // A block that takes everything that is left of a complete input needs no row scan of its own: pass 1 hands its spans out in
    // order, each wave counts the rows that start in its span and learns its first row's number from the spans before (a full
    // read of the input less, -1.8 ms per 10 GB).  The record matrix is then sized from the bytes per row seen so far (the
    // input's first 64 KiB, or the block before) and a quarter more; a block that turns out to hold more rows is parsed again
    // the ordinary way.
    fold = !(hint && hint->max_rows != 0xFFFFFFFFu) && !c->no_fold && !c->oom_seen && !test_hook("ZDW_P1_NO_FOLD");
    fold_rows_cap = 0;
    for (;;) {
    if (c->in_stream) { int rcs = stream_snapshot(c, need); if (rcs) return rcs; }
    hi = c->hi;
    if (windowed && reach < (double)(c->hi - lo)) hi = lo + (uint64_t)reach;
    if (fold) {
        const double bpr = c->bytes_per_row > 0.0 ? c->bytes_per_row : (double)c->avg_row;
        fold = !c->input_open && bpr >= 1.0 && hi > lo;
        if (fold) {
            const double cap = (double)(hi - lo) / bpr * 1.25 + 4096.0;
            fold = cap < 4.0e9 && cap * (double)c->rec_stride < 6.0e10;      // (an estimate that asks for a huge matrix: count first)
            fold_rows_cap = (uint64_t)cap;
        }
    }
    for (;;) {
    if (hi > lo) {
        StageScope ts(c, ZDW_STAGE_SCAN_ROWS);
        // span size by block size: visits cost a few microseconds of start-up latency each, so large blocks take
        // 64 KiB spans; small ones keep 16 KiB so that there are enough spans for every resident wave
        // (measured in round 4, the benchmark's table: 1 GB best at 64 KiB spans, 4 GB at 128 KiB, 10 GB at 256 KiB -- the largest span that
        // still leaves about 20 000 of them, three to each resident wave; until then: 80 000 spans, at most 64 KiB: 4 % slower at 10 GB)
        span_shift = SPAN_SHIFT_MIN;
        while (span_shift < SPAN_SHIFT_MAX && ((hi - lo) >> (span_shift + 1)) >= 20000) ++span_shift;
        {
            // ... and by row length: a visit skips what is left of the row its span starts in and follows its last row
            // past the span's end, i.e. tokenises about one row more than it owns.  Spans of at least 8 average rows
            // (sampled at the start of the input) keep that below an eighth, as long as 16 K spans remain for the resident
            // waves.  (2086 columns, 19 KB rows, 4 GB: spans of 32 / 128 / 256 / 512 KiB: pass 1 12.4 / 7.7 / 7.3 / 7.8 ms.)
            const uint64_t avg_row = c->avg_row ? c->avg_row : 1;      // sampled once per input (after_input)
            while (span_shift < 20 && (1ull << span_shift) < 8 * avg_row && ((hi - lo) >> (span_shift + 1)) >= 16384) ++span_shift;
        }
        if (const char *e = test_hook("ZDW_SPAN_SHIFT")) span_shift = std::min(20u, std::max(12u, (uint32_t)atoi(e)));      // tests / tuning
        span0 = lo >> span_shift;
        nspans = (uint32_t)(((hi + (1ull << span_shift) - 1) >> span_shift) - span0);
        if (fold) { total_rows = fold_rows_cap; break; }          // (what pass 1 may find; it says how many there were)
        HIP_TRY(c, c->span_starts.ensure((nspans + 1ull) * 4));
        HIP_TRY(c, c->span_rowbase.ensure((nspans + 2ull) * 4));
        HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(nspans + 1ull)));
        P1Status init{};
        init.bad_row = 0xFFFFFFFFu; init.split_pos = c->n;
        HIP_TRY(c, hipMemcpyAsync(c->status.p, &init, sizeof init, hipMemcpyHostToDevice, s));
        launch_count_row_starts(c->d_in, c->n, lo, hi, span0, nspans, span_shift, c->span_starts.as<uint32_t>(), c->status.as<P1Status>(), s);
        launch_scan_u32_to_u32(c->span_starts.as<uint32_t>(), c->span_rowbase.as<uint32_t>(), nspans, c->scan_tmp.p,
                               c->d_scalar.as<uint64_t>(), s);
        HIP_TRY(c, hipMemcpyAsync(&total_rows, c->d_scalar.p, 8, hipMemcpyDeviceToHost, s));
        P1Status st0{};
        HIP_TRY(c, hipMemcpyAsync(&st0, c->status.p, sizeof st0, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (st0.nul_seen) return fail(c, ZDW_UNKNOWN_ERROR, "input contains a NUL byte: not representable in ZDW (C strings)");
    }
    if (hi < c->hi && total_rows <= hint->max_rows) { hi = (c->hi - lo) / 2 < hi - lo ? c->hi : lo + 2 * (hi - lo); continue; }
    break;
    }
    if (c->input_open && !(hint && hint->max_rows != 0xFFFFFFFFu && hint->max_rows < total_rows)) {
        need = need == ~0ull ? need : c->n + std::max<uint64_t>(c->n / 4, 64ull << 20);
        continue;
    }
    break;
    }
    return ZDW_OK;
}

int Pass1Job::buffers()
{
    c->next_lo = c->n;
    if (total_rows >= 0xFFFFFFFFull) return fail(c, ZDW_UNKNOWN_ERROR, "more than 2^32-2 rows in one block");
    lap.mark("scan");

    row_limit = 0xFFFFFFFFu; partial = 0;
    tripped = false;
    if (hint && hint->max_rows != 0xFFFFFFFFu && hint->max_rows < total_rows) {
        tripped = true; row_limit = hint->max_rows; partial = hint->partial_fields;
    }
    nrows = tripped ? row_limit : (uint32_t)total_rows;

    // ---- pass 1 proper (also runs for zero rows when a partial row pollutes the block) ----
    // column facts and ranges live in per-wave LDS when they fit there (and in the packed 14 + 16 bit form)
    lds_ranges = pass1_lds_bytes(c->ncols, c->nnum, true) - pass1_lds_bytes(c->ncols, c->nnum, false) <= 40 * 1024 &&
                            c->nnum < 16384 && c->rec_stride < 65536;
    nslots = first_table(nrows);
    HIP_TRY(c, c->col_min.ensure(c->ncols * 8 + 8)); HIP_TRY(c, c->col_max.ensure(c->ncols * 8 + 8)); HIP_TRY(c, c->col_seen.ensure(c->ncols + 8));
    if (c->model_max_slots && !c->model_rows_prev && tripped) {
        // The first window of an input whose blocks the memory-limit model ends: the next block's window is a quarter larger (room for that and some more).
        // Room for that now saves freeing and allocating tens of GB between the first two blocks (seen to take half a second).
        (void)c->rec.ensure((uint64_t)nrows * c->rec_stride / 5 * 8 + 64);
    }
    for (;;) {
        const hipError_t e = c->rec.ensure((uint64_t)nrows * c->rec_stride + 64);      // slack: 8-byte reads of 4-byte fields, 16-byte staging reads
        if (e == hipSuccess) break;
        if (e == hipErrorOutOfMemory && fold) { c->no_fold = true; return P1_COUNT_FIRST; }
        if (e != hipErrorOutOfMemory || !fewer_rows())
            return fail(c, e == hipErrorOutOfMemory ? ZDW_OUT_OF_MEMORY : ZDW_UNKNOWN_ERROR, "record matrix of %llu bytes: %s",
                        (unsigned long long)nrows * c->rec_stride, hipGetErrorString(e));
    }
    lap.mark("rec_alloc");
    const uint64_t blk_bytes = hi > lo ? hi - lo : 0;
    long_cap = (uint32_t)std::min<uint64_t>(blk_bytes / (FIRST_ROW_CAP - 1) + 2, 1u << 26);
    HIP_TRY(c, c->long_rows.ensure((uint64_t)long_cap * sizeof(LongRow)));
    return ZDW_OK;
}

int Pass1Job::parse()
{
    st = P1Status{};
    bool dq_grown = false;
    int reruns = 0;                  // re-runs that reset `attempt`: bounded on their own
    for (int attempt = 0;; ++attempt) {
        {   // the table and the unique ids: the table's load limit plus what waves can leave unused in their chunks (holes)
            const uint64_t cap = std::min<uint64_t>(nslots / 8 * 5 + 64ull * 8192 + nslots / 16, 0xFFFFFFF0ull);
            hipError_t e = nslots > (1ull << 31) ? hipErrorOutOfMemory : c->slots.ensure(nslots * 32);
            if (e == hipSuccess) e = c->uniq_pos.ensure(cap * 8);
            if (e == hipSuccess) e = c->uniq_len.ensure(cap * 4);
            if (e == hipSuccess) {
                // optional (without it the sort and the blob go back to the image for every string): not under a memory budget
                // (zdw_opts.max_device_bytes), where every byte decides how many rows a block can hold
                if (c->budget.cap || c->uniq_key.ensure(cap * 24) != hipSuccess) c->uniq_key.release();
            }
            if (e == hipErrorOutOfMemory && fold) { c->no_fold = true; return P1_COUNT_FIRST; }
            if (e == hipErrorOutOfMemory) {
                // no room for a table of this block's uniques: fewer rows, fewer uniques (and a smaller record matrix)
                if (!fewer_rows()) return fail(c, ZDW_OUT_OF_MEMORY, "no room for a dictionary table of %llu slots; split the block", (unsigned long long)nslots);
                c->slots.release(); c->uniq_pos.release(); c->uniq_len.release(); c->uniq_key.release(); c->rec.release();
                while (c->rec.ensure((uint64_t)nrows * c->rec_stride + 64) != hipSuccess)
                    if (!fewer_rows()) return fail(c, ZDW_OUT_OF_MEMORY, "no room for the record matrix of a single row; split the block");
                nslots = std::min(nslots, first_table(nrows));
                if (++reruns > 40) return fail(c, ZDW_OUT_OF_MEMORY, "no room for this block's buffers");
                attempt = -1;
                continue;
            }
            if (e != hipSuccess) return fail(c, ZDW_UNKNOWN_ERROR, "dictionary table: %s", hipGetErrorString(e));
            c->id_cap = (uint32_t)cap;
        }
        lap.mark("table_alloc");
        HIP_TRY(c, hipMemsetAsync(c->slots.p, 0, nslots * 32, s));
        HIP_TRY(c, hipMemsetAsync(c->col_min.p, 0xFF, c->ncols * 8 + 8, s));
        HIP_TRY(c, hipMemsetAsync(c->col_max.p, 0, c->ncols * 8 + 8, s));
        HIP_TRY(c, hipMemsetAsync(c->col_seen.p, 0, c->ncols + 8, s));
        P1Status init{};
        init.bad_row = 0xFFFFFFFFu; init.split_pos = c->n;
        HIP_TRY(c, hipMemcpyAsync(c->status.p, &init, sizeof init, hipMemcpyHostToDevice, s));
        c->nslots = nslots;
        bool defer_used = false, rows_used = false;
        uint32_t dq_cap_used = 0;
        if (hi > lo && total_rows) {
            StageScope ts(c, ZDW_STAGE_PASS1);
            Pass1Args a{};
            a.in = c->d_in; a.n = c->n; a.lo = lo; a.hi = hi; a.span0 = span0; a.nspans = nspans; a.span_shift = span_shift;
            a.span_rowbase = c->span_rowbase.as<uint32_t>();
            a.extra = c->extra.as<uint8_t>();
            a.ncols = c->ncols; a.col_kind = c->d_kind.as<uint8_t>(); a.col_recoff = c->d_recoff.as<uint32_t>();
            a.col_numidx = c->d_numidx.as<int32_t>(); a.nnum = c->nnum; a.use_lds_ranges = lds_ranges ? 1u : 0u;
            a.rec_stride = c->rec_stride; a.rec = c->rec.as<uint8_t>();
            a.row_limit = row_limit; a.partial_fields = partial; a.trim = c->trim ? 1u : 0u;
            if (fold) {
                HIP_TRY(c, c->span_state.ensure((uint64_t)nspans * 8 + 64));
                HIP_TRY(c, hipMemsetAsync(c->span_state.p, 0, (uint64_t)nspans * 8, s));
                a.span_state = c->span_state.as<uint64_t>();
                a.row_limit = (uint32_t)std::min<uint64_t>(fold_rows_cap, 0xFFFFFFFEull);      // rows the matrix has room for
            }
            a.slots = c->slots.as<uint64_t>(); a.slot_mask = nslots - 1;
            a.uniq_pos = c->uniq_pos.as<uint64_t>(); a.uniq_len = c->uniq_len.as<uint32_t>(); a.id_cap = c->id_cap; a.uniq_key = c->uniq_key.as<uint64_t>();
            a.max_entries = (uint32_t)std::min<uint64_t>(nslots / 8 * 5, 0xFFFFFFF0ull);
            a.col_min = c->col_min.as<uint64_t>(); a.col_max = c->col_max.as<uint64_t>(); a.col_seen = c->col_seen.as<uint8_t>();
#ifdef ZDW_EXPERIMENTS
            if (const char *dbg = experiment("ZDW_DEBUG_SKIP")) a.debug_skip = (uint32_t)atoi(dbg);
#endif
            // the memory-limit model asks in which order the strings are first met: every occurrence takes part in a minimum
            a.track_first = (c->model_max_slots || c->track_first_opt) ? 1u : 0u;
            if (const char *e = test_hook("ZDW_P1_TRACK_FIRST")) a.track_first = atoi(e) != 0;      // tests
            if (a.track_first) HIP_TRY(c, hipMemsetAsync(c->uniq_pos.p, 0xFF, (uint64_t)c->id_cap * 8, s));
            c->last_track = a.track_first != 0;
            a.dense_boundaries = c->dense_boundaries ? 1u : 0u;
            if (const char *e = test_hook("ZDW_P1_LARGE_RING")) a.dense_boundaries = atoi(e) != 0;      // tests
            a.skip_empty = c->skip_empty ? 1u : 0u;
            if (const char *e = test_hook("ZDW_P1_SKIP_EMPTY")) a.skip_empty = atoi(e) != 0;      // tests
            a.rebase_at = a.skip_empty ? 1u << 23 : 1u << 30;              // (descriptors of that form hold 24 bits of position)
            if (const char *e = test_hook("ZDW_P1_REBASE_AT")) a.rebase_at = std::min(a.rebase_at, std::max(16384u, (uint32_t)strtoul(e, nullptr, 10)));   // tests
            if ((a.dense_boundaries || a.skip_empty) && nrows) {
                // mostly empty fields: pass 1 skips every empty field.  Either the record matrix is zeroed in one streaming write, or
                // (the form for such tables) a map of a bit per field says where a record was written and pass 2 looks nowhere
                // else: rows x columns / 8 bytes to clear instead of rows x ~6 x columns.
                c->nz_stride = ((c->ncols + 31u) / 32u + 3u) & ~3u;
                const uint64_t map_bytes = (uint64_t)nrows * c->nz_stride * 4u;
                bool use_map = a.skip_empty && map_bytes <= (16ull << 30) && !test_hook("ZDW_P1_NO_NZ");
                if (use_map && c->nz.ensure(map_bytes + 64) != hipSuccess) use_map = false;
                if (use_map) {
                    HIP_TRY(c, hipMemsetAsync(c->nz.p, 0, map_bytes, s));
                    a.nz = c->nz.as<uint32_t>(); a.nz_stride = c->nz_stride;
                } else {
                    HIP_TRY(c, hipMemsetAsync(c->rec.p, 0, (uint64_t)nrows * c->rec_stride, s));
                }
                c->nz_mode = use_map; c->rec_dense = !use_map; c->nrows_written = nrows;
                a.rec_zeroed = 1;
            }
            a.st = c->status.as<P1Status>(); a.long_rows = c->long_rows.as<LongRow>(); a.long_cap = long_cap;
            // Fields that need the general form (long strings, signed / padded numbers, contended slots) are
            // queued by the main kernel (16 bytes each) and handled by a second one.  The queue starts at about 1 %
            // of the block's fields; a pass that needs more says how many and is run again with that (the context
            // remembers), up to one queue byte per input byte -- beyond that, or with a schema whose tables do
            // not fit LDS, the all-in-one kernel runs instead.  (10 GB x 200 columns with words of 4..32 / 4..96
            // characters: queue 19 / 27 ms, all-in-one 34 / 42 ms.)
            const bool defer = lds_ranges && !c->p1_all_in_one && !test_hook("ZDW_P1_ALL_IN_ONE");
            if (defer) {
                uint64_t dq_div = 256;
                if (const char *e = experiment("ZDW_DQ_DIV")) dq_div = std::max<uint64_t>(8, strtoull(e, nullptr, 10));      // experiments
                // (what earlier passes of this context queued per input byte, with a quarter to spare, when that is more)
                const uint64_t by_rate = (uint64_t)((double)(hi - lo) * c->dq_rate * 1.25);
                const uint64_t cap = std::min<uint64_t>(std::max<uint64_t>(std::max<uint64_t>((hi - lo) / dq_div + 65536, by_rate), c->dq_learned), 1ull << 31);
                HIP_TRY(c, c->deferred_q.ensure(cap * 16));
                a.dq = c->deferred_q.as<uint64_t>(); a.dq_cap = (uint32_t)cap;
                defer_used = true; dq_cap_used = (uint32_t)cap;
            }
            // Two forms of the kernel (the same records, ranges, table and status): lanes take consecutive FIELDS (k_pass1.hip) or a lane
            // walks its own ROWS (k_rows.hip: a streaming row index first, then wave-uniform columns).  The second wants the
            // general-form queue, rows that are long against its start-up costs, and none of the other forms' special cases.
            const bool rows_can = defer && !a.trim && !a.skip_empty && !a.dense_boundaries && pass1_rows_lds_bytes(c->ncols, c->nnum, 1) <= 160u * 1024u;
            bool rows_form = false;
            // (tests: 0 never, 1 where it is expected to pay, 2 wherever it can run)
            if (const char *e = test_hook("ZDW_P1_ROWS")) rows_form = rows_can && !c->no_rows_form && (atoi(e) >= 2 || (atoi(e) == 1 && c->ncols >= 16 && c->avg_row >= 256));
            rows_used = rows_form;
            if (rows_form) {
                const uint64_t rx0 = lo >> 16, rxn = ((hi + 65535) >> 16) - rx0;              // the row index's spans: 64 KiB
                const uint64_t rows_cap = fold ? fold_rows_cap : total_rows;
                HIP_TRY(c, c->span_state.ensure(rxn * 8 + 64));
                HIP_TRY(c, c->rx_bsmap.ensure(rxn * 4 + 64));                                 // (a byte per 16 KiB)
                HIP_TRY(c, c->rowidx.ensure((rows_cap + 4) * 16));
                HIP_TRY(c, hipMemsetAsync(c->span_state.p, 0, rxn * 8, s));
                RowIndexArgs ri{};
                ri.in = c->d_in; ri.n = c->n; ri.lo = lo; ri.hi = hi; ri.span0 = rx0; ri.nspans = (uint32_t)rxn;
                ri.span_state = c->span_state.as<uint64_t>(); ri.rowidx = c->rowidx.as<uint64_t>(); ri.rows_cap = rows_cap;
                ri.bsmap = c->rx_bsmap.as<uint8_t>(); ri.st = a.st;
                launch_row_index(ri, s);
                RowsArgs ra{};
                ra.rowidx = c->rowidx.as<uint64_t>(); ra.bsmap = c->rx_bsmap.as<uint8_t>(); ra.bs_span0 = rx0 * 4;
                ra.rows_per_lane = 4;
                if (const char *e = test_hook("ZDW_P1_ROWS_PER_LANE")) ra.rows_per_lane = std::max(1, atoi(e));
                if (const char *e = experiment("ZDW_P1_ROWS_FIND")) {
                    // measurement only: the field form fills the table first, then the row form runs over the same input looking up only
                    if (atoi(e)) { Pass1Args b = a; launch_pass1(b, s); ra.find_only = 1; }
                }
                a.span_state = nullptr;
                launch_pass1_rows(a, ra, s);
                c->last_span_shift = 16; c->last_span0 = rx0; c->last_fold = true;      // (the row index leaves the look-back's words: locate_occurrence)
            } else {
                launch_pass1(a, s);
                c->last_span_shift = span_shift; c->last_span0 = span0; c->last_fold = fold;
            }
            ++c->pass1_launches;
            // the queued general-form fields: how many there are, and whether the queue or the table overflowed (then the pass is run
            // again and this launch does nothing), the kernel reads on the device -- no host round trip between the two launches
            if (defer) launch_pass1_deferred(a, 0xFFFFFFFFu, s);
        }
        HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (rows_used && st.fold_abort) {
            // (the row index never came together: its rows mean nothing -- the field form, which counts for itself)
            c->no_rows_form = true;
            if (++reruns > 40) return fail(c, ZDW_UNKNOWN_ERROR, "pass 1: the row index keeps failing");
            --attempt;
            continue;
        }
        if (defer_used) {
            if (st.deferred_lost) {
                if (experiment("ZDW_DEBUG_RETRY")) fprintf(stderr, "[pass1] deferred queue of %u entries lost fields (%u queued): again\n", dq_cap_used, st.deferred);
                const uint64_t need = (uint64_t)st.deferred + st.deferred / 8 + 65536;       // (a re-run hands out its chunks differently)
                if (!dq_grown && need * 16 <= (hi - lo) + (64ull << 20) && need > dq_cap_used && need < (1ull << 31) && !st.overflow) {
                    c->dq_learned = need;
                    dq_grown = true;              // once per block: a second loss goes to the all-in-one kernel
                } else {
                    c->p1_all_in_one = true;      // this block and the context's later ones
                }
                --attempt;
                continue;
            }
            if (hi > lo) c->dq_rate = std::max(c->dq_rate, (double)st.deferred / (double)(hi - lo));
        }
#ifdef ZDW_EXPERIMENTS
        if (experiment("ZDW_P1_ROWS_DBG")) { fprintf(stderr, "[rows dbg cycles]"); for (int i = 0; i < 12; ++i) fprintf(stderr, " %llu", st.dbg[i]); fprintf(stderr, "\n"); }
#endif
        lap.mark("pass1_kernels");
        hipError_t le = hipGetLastError();      // (allocation failures handled above have cleared theirs: DevBuf::ensure)
        if (le != hipSuccess) return fail(c, ZDW_UNKNOWN_ERROR, "pass 1 launch: %s", hipGetErrorString(le));
        if (st.nul_seen == 2) return fail(c, ZDW_UNKNOWN_ERROR, "a single field of 1 GiB or more is not supported");
        if (fold) {
            if (st.nul_seen == 1) return fail(c, ZDW_UNKNOWN_ERROR, "input contains a NUL byte: not representable in ZDW (C strings)");
            if (st.fold_abort || (!st.overflow && (st.rows_total == 0 || st.rows_total - 1 > fold_rows_cap))) {
                // more rows than the estimate left room for (or the count never came together): the ordinary way
                if (experiment("ZDW_DEBUG_RETRY")) fprintf(stderr, "[pass1] %llu rows where %llu were expected: again, counting first\n", (unsigned long long)(st.rows_total ? st.rows_total - 1 : 0), (unsigned long long)fold_rows_cap);
                c->no_fold = true;
                return P1_COUNT_FIRST;
            }
            if (!st.overflow) { total_rows = st.rows_total - 1; nrows = (uint32_t)total_rows; }
        }
        if (!st.overflow) break;
        if (experiment("ZDW_DEBUG_RETRY")) fprintf(stderr, "[pass1] table of %llu slots overflowed (%u entries, %u ids of %u): again\n", (unsigned long long)nslots, st.entries, st.id_next, c->id_cap);
        if (c->model_max_slots && nslots >= c->model_max_slots && nrows > 1) {
            // The memory-limit model ends a block long before its dictionary needs a larger table than this: these rows hold
            // more strings than any block may -- look at half of them instead of growing the table.
            if (!fewer_rows()) {
                // (pass 1 was counting its own rows, so there is no row count to halve: count first, the ordinary way)
                if (fold) { c->no_fold = true; return P1_COUNT_FIRST; }
                return fail(c, ZDW_OUT_OF_MEMORY, "dictionary table of %llu slots overflows on a single row", (unsigned long long)nslots);
            }
            if (++reruns > 40) return fail(c, ZDW_OUT_OF_MEMORY, "dictionary table keeps overflowing");
            attempt = -1;
            continue;
        }
        nslots <<= 2;                       // table too small for this block's uniques: grow and redo the pass
        if (c->model_max_slots && nslots > c->model_max_slots) nslots = c->model_max_slots;
        if (attempt > 12) return fail(c, ZDW_OUT_OF_MEMORY, "dictionary table keeps overflowing");
    }
    if (experiment("ZDW_DEBUG_RETRY")) { fprintf(stderr, "[pass1] %u rows, table %llu slots, %u entries; hipMalloc/hipFree so far %.1f ms\n", nrows, (unsigned long long)nslots, st.entries, g_alloc_ms); }
    return ZDW_OK;
}

int Pass1Job::results()
{
    c->nrows = nrows;
    if (c->nz_mode) c->nrows_written = nrows;      // (rows beyond the block's were not written)
    c->is_last = !tripped;
    c->partial_taken = tripped ? partial : 0;
    res->table_slots = nslots;
    c->slots_learned = nslots; c->rows_learned = nrows;
    if (nrows) c->entries_per_row = (double)st.entries / (double)nrows;      // (rows beyond a split add theirs: an upper bound)
    if (nrows && (tripped ? st.split_pos : hi) > lo) c->bytes_per_row = (double)((tripped ? st.split_pos : hi) - lo) / (double)nrows;
    res->dict_entries = st.entries;

    if (st.bad_row != 0xFFFFFFFFu && st.bad_row <= row_limit) {          // ConvertToZDW.cpp:810-812
        res->bad_row = st.bad_row + 1;
        if (!c->defer_bad_row) return fail(c, ZDW_WRONG_NUM_OF_COLUMNS_ON_A_ROW, "Row %u had the problem", res->bad_row);
    }

    // ---- m_LongestLine: replay the reference's buffer growth on rows that could have triggered it ----
    {
        c->long_list.clear();
        uint32_t cnt = st.long_count;
        if (cnt > long_cap) cnt = long_cap;
        if (cnt) {
            c->long_list.resize(cnt);
            HIP_TRY(c, hipMemcpy(c->long_list.data(), c->long_rows.p, cnt * sizeof(LongRow), hipMemcpyDeviceToHost));
            std::sort(c->long_list.begin(), c->long_list.end(), [](const LongRow &x, const LongRow &y) { return x.start < y.start; });
        }
        c->long_read_end = tripped ? st.split_pos : c->n;   // pass 1 reads rows up to and including the one at the split
        c->long_tail = !tripped && c->n > hi;               // an unterminated tail is read (and dropped) by pass 1 as well
        int rc_lr = replay_long_rows(c, &c->row_cap);
        if (rc_lr) return rc_lr;
    }
    res->longest_line = c->row_cap;
    res->num_rows = nrows;
    res->is_last_block = tripped ? 0 : 1;
    c->next_lo = tripped ? st.split_pos : c->n;
    res->bytes_consumed = c->next_lo;

    // column ranges to the host (the multi-GPU path may merge them before they are used)
    c->h_min.assign(c->ncols, 0); c->h_max.assign(c->ncols, 0); c->h_seen.assign(c->ncols, 0);
    if (c->ncols) {
        HIP_TRY(c, hipMemcpy(c->h_min.data(), c->col_min.p, c->ncols * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_max.data(), c->col_max.p, c->ncols * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(c->h_seen.data(), c->col_seen.p, c->ncols, hipMemcpyDeviceToHost));
    }
    c->p1_done = true;
    c->st_host = st; c->st_known = true;
    lap.mark("tail");
    lap.done("pass1_impl");
    if (tripped && nrows == 0) return fail(c, ZDW_OUT_OF_MEMORY, "block limit leaves no room for a single row");
    return ZDW_OK;
}

static int pass1_impl(zdw_ctx *c, const zdw_split_hint *hint, zdw_pass1_result *res)
{
    for (int round = 0; round < 3; ++round) {
        Pass1Job job{c, hint, res};
        int rc = job.window_and_rows();
        if (rc == ZDW_OK) rc = job.buffers();
        if (rc == ZDW_OK) rc = job.parse();
        if (rc == ZDW_OK) rc = job.results();
        if (rc != P1_COUNT_FIRST) return rc;
        c->no_fold = true;          // again, with the rows counted first
    }
    return fail(c, ZDW_UNKNOWN_ERROR, "pass 1 keeps asking for a row count");
}

int zdw_block_dict_device(zdw_ctx *c, const void **d_section, uint64_t *section_len, uint8_t *index_size)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    int rc = ensure_dict(c);
    if (rc) return rc;
    if (d_section) *d_section = c->dict_sec.as<uint8_t>() + c->dict_sec_off;
    if (section_len) *section_len = c->dict_sec_len;
    if (index_size) *index_size = c->nuniq ? c->index_size : 0;
    return ZDW_OK;
}

int zdw_block_dict(zdw_ctx *c, const uint8_t **section, uint64_t *section_len, uint8_t *index_size)
{
    const void *d = nullptr;
    uint64_t len = 0;
    int rc = zdw_block_dict_device(c, &d, &len, index_size);
    if (rc) return rc;
    HIP_TRY(c, c->h_dict.ensure(len));
    {
        StageScope ts(c, ZDW_STAGE_D2H);
        HIP_TRY(c, hipMemcpyAsync(c->h_dict.p, d, len, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (section) *section = static_cast<const uint8_t *>(c->h_dict.p);
    if (section_len) *section_len = len;
    return ZDW_OK;
}

int zdw_block_stats(zdw_ctx *c, uint8_t *col_size, uint64_t *col_base, uint32_t *used)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    int rc = ensure_stats(c);
    if (rc) return rc;
    if (col_size && c->ncols) memcpy(col_size, c->col_size.data(), c->ncols);
    if (col_base) for (size_t u = 0; u < c->used.size(); ++u) col_base[u] = c->col_base[c->used[u]];
    if (used) *used = (uint32_t)c->used.size();
    return ZDW_OK;
}

int zdw_block_pass2_device(zdw_ctx *c, const void **d_rows, uint64_t *rows_len)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    int rc = run_pass2(c);
    if (rc) return rc;
    if (d_rows) *d_rows = c->rows_out.p;
    if (rows_len) *rows_len = c->rows_len;
    return ZDW_OK;
}

int zdw_block_pass2(zdw_ctx *c, const uint8_t **rows, uint64_t *rows_len)
{
    const void *d = nullptr;
    uint64_t len = 0;
    int rc = zdw_block_pass2_device(c, &d, &len);
    if (rc) return rc;
    HIP_TRY(c, c->h_rows.ensure(len + 1));
    if (len) {
        StageScope ts(c, ZDW_STAGE_D2H);
        HIP_TRY(c, hipMemcpyAsync(c->h_rows.p, d, len, hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (rows) *rows = static_cast<const uint8_t *>(c->h_rows.p);
    if (rows_len) *rows_len = len;
    return ZDW_OK;
}

namespace {
constexpr uint64_t PIECE_BYTES = 32ull << 20;

// device bytes [d, d + len) to `write`, double-buffered through pinned memory
int stream_out(zdw_ctx *c, const uint8_t *d, uint64_t len, zdw_write_fn write, void *user)
{
    if (!len) return ZDW_OK;
    hipStream_t s = c->stream;
    const uint64_t piece = std::min<uint64_t>(PIECE_BYTES, len);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(c, c->h_piece[i].ensure(piece));
        if (!c->piece_ev[i]) HIP_TRY(c, hipEventCreateWithFlags(&c->piece_ev[i], hipEventDisableTiming));
    }
    StageScope ts(c, ZDW_STAGE_D2H);
    const uint64_t n = (len + piece - 1) / piece;
    auto issue = [&](uint64_t k) -> hipError_t {
        const uint64_t off = k * piece, m = std::min<uint64_t>(piece, len - off);
        hipError_t e = hipMemcpyAsync(c->h_piece[k & 1].p, d + off, m, hipMemcpyDeviceToHost, s);
        if (e != hipSuccess) return e;
        return hipEventRecord(c->piece_ev[k & 1], s);
    };
    HIP_TRY(c, issue(0));
    for (uint64_t k = 0; k < n; ++k) {
        if (k + 1 < n) HIP_TRY(c, issue(k + 1));
        HIP_TRY(c, hipEventSynchronize(c->piece_ev[k & 1]));
        const uint64_t off = k * piece, m = std::min<uint64_t>(piece, len - off);
        if (write(user, static_cast<const uint8_t *>(c->h_piece[k & 1].p), m) != 0) {
            (void)hipStreamSynchronize(s);
            return fail(c, ZDW_UNKNOWN_ERROR, "the writer refused %llu bytes of output", (unsigned long long)m);
        }
    }
    return ZDW_OK;
}
}  // namespace

int zdw_block_dict_write(zdw_ctx *c, zdw_write_fn write, void *user, uint64_t *section_len, uint8_t *index_size)
{
    if (!c || !write) return ZDW_UNKNOWN_ERROR;
    const void *d = nullptr;
    uint64_t len = 0;
    int rc = zdw_block_dict_device(c, &d, &len, index_size);
    if (rc) return rc;
    if (section_len) *section_len = len;
    return stream_out(c, static_cast<const uint8_t *>(d), len, write, user);
}

int zdw_block_pass2_write(zdw_ctx *c, zdw_write_fn write, void *user, uint64_t *rows_len)
{
    if (!c || !write) return ZDW_UNKNOWN_ERROR;
    const void *d = nullptr;
    uint64_t len = 0;
    int rc = zdw_block_pass2_device(c, &d, &len);
    if (rc) return rc;
    if (rows_len) *rows_len = len;
    return stream_out(c, static_cast<const uint8_t *>(d), len, write, user);
}

int zdw_block_next(zdw_ctx *c)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    c->lo = c->next_lo;
    reset_block(c);
    if (c->in_stream) {                     // images the stream has outgrown: no block reads them any more
        std::vector<void *> old;
        { std::lock_guard<std::mutex> g(c->in_stream->m); old.swap(c->in_stream->retired); }
        if (!old.empty()) { (void)hipStreamSynchronize(c->stream); if (c->stream2) (void)hipStreamSynchronize(c->stream2); }
        for (void *p : old) (void)hipFree(p);
    }
    return ZDW_OK;
}

// ---- multi-GPU shard exchange ---------------------------------------------------------------------
int zdw_shard_export_ranges(zdw_ctx *c, uint64_t *col_min, uint64_t *col_max, uint8_t *col_seen)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    for (uint32_t k = 0; k < c->ncols; ++k) {
        if (col_min) col_min[k] = c->h_min[k];
        if (col_max) col_max[k] = c->h_max[k];
        if (col_seen) col_seen[k] = c->h_seen[k];
    }
    return ZDW_OK;
}

int zdw_shard_import_ranges(zdw_ctx *c, const uint64_t *col_min, const uint64_t *col_max, const uint8_t *col_seen)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    for (uint32_t k = 0; k < c->ncols; ++k) {
        if (col_min) c->h_min[k] = col_min[k];
        if (col_max) c->h_max[k] = col_max[k];
        if (col_seen) c->h_seen[k] = col_seen[k];
    }
    c->stats_done = false;
    invalidate_pass2(c);
    return ZDW_OK;
}

int zdw_shard_export_uniques(zdw_ctx *c, const void **d_lens, const void **d_bytes, uint64_t *count, uint64_t *nbytes)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    // the uniques as pass 1 left them, by id; ids with length 0 are holes and travel as such (no bytes)
    uint32_t ids = 0;
    HIP_TRY(c, hipMemcpy(&ids, &c->status.as<P1Status>()->id_next, 4, hipMemcpyDeviceToHost));
    if (ids > c->id_cap) return fail(c, ZDW_UNKNOWN_ERROR, "unique id counter (%u) beyond its table (%u)", ids, c->id_cap);
    const uint64_t u = ids;
    HIP_TRY(c, c->lenp1.ensure((u + 1) * 4)); HIP_TRY(c, c->off_excl.ensure((u + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(u + 1)));
    launch_dict_lengths(nullptr, c->uniq_len.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);      // identity order
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->off_excl.as<uint64_t>() + u, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, c->exp_bytes.ensure(total + 16));
    launch_export_strings(c->d_in, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), nullptr, c->off_excl.as<uint64_t>(), u,
                          c->exp_bytes.as<uint8_t>(), nullptr, s, c->uniq_key.as<uint64_t>());
    HIP_TRY(c, hipStreamSynchronize(s));
    if (d_lens) *d_lens = c->uniq_len.p;
    if (d_bytes) *d_bytes = c->exp_bytes.p;
    if (count) *count = u;
    if (nbytes) *nbytes = total;
    return ZDW_OK;
}

int zdw_shard_import_uniques(zdw_ctx *c, const void *d_lens, const void *d_bytes, uint64_t count, uint64_t nbytes)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    if (!count) return ZDW_OK;
    if (nbytes >= EXTRA_BASE) return fail(c, ZDW_OUT_OF_MEMORY, "imported strings exceed 2^39 bytes");
    // own copies: the caller's buffers may be reused while this block is alive
    HIP_TRY(c, c->extra.ensure(nbytes + 16));
    HIP_TRY(c, c->extra_lens.ensure((count + 1) * 4));
    HIP_TRY(c, c->extra_offs.ensure((count + 2) * 8));
    HIP_TRY(c, c->lenp1.ensure((count + 1) * 4));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(count + 1)));
    HIP_TRY(c, hipMemcpyAsync(c->extra.p, d_bytes, nbytes, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->extra_lens.p, d_lens, count * 4, hipMemcpyDeviceToDevice, s));
    launch_dict_lengths(nullptr, c->extra_lens.as<uint32_t>(), count, c->lenp1.as<uint32_t>(), s);      // identity order
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->extra_offs.as<uint64_t>(), count, c->scan_tmp.p, nullptr, s);
    {   // the lengths must describe exactly the bytes that came with them, or the insert kernel would walk off the buffer
        uint64_t total = 0;
        HIP_TRY(c, hipMemcpyAsync(&total, c->extra_offs.as<uint64_t>() + count, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (total != nbytes)
            return fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_import_uniques: lengths add up to %llu bytes, %llu were passed",
                        (unsigned long long)total, (unsigned long long)nbytes);
    }
    c->st_known = false;
    launch_import_strings(c->d_in, c->n, c->extra.as<uint8_t>(), c->extra_lens.as<uint32_t>(), c->extra_offs.as<uint64_t>(), count,
                          c->slots.as<uint64_t>(), c->nslots - 1,
                          (uint32_t)std::min<uint64_t>(c->nslots - c->nslots / 8, 0xFFFFFFF0ull), c->uniq_pos.as<uint64_t>(),
                          c->uniq_len.as<uint32_t>(), c->id_cap, c->trim ? 1u : 0u, c->hi, c->status.as<P1Status>(), nullptr, s, c->uniq_key.as<uint64_t>());
    P1Status st{};
    HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->st_host = st; c->st_known = true;
    if (st.overflow)
        return fail(c, ZDW_OUT_OF_MEMORY, "dictionary table (%llu slots) too small for the merged dictionary: raise zdw_opts.dict_slots",
                    (unsigned long long)c->nslots);
    c->extra_bytes = nbytes;
    c->dict_size_done = c->dict_done = c->stats_done = false;
    invalidate_pass2(c);
    return ZDW_OK;
}

int zdw_shard_line_caps(zdw_ctx *c, uint32_t *caps)
{
    if (!c || !caps || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    for (uint32_t k = 0; k < ZDW_LINE_CAPS; ++k) {
        const uint64_t start = (uint64_t)FIRST_ROW_CAP << k;
        uint32_t cap = start > 0x80000000ull ? 0x80000000u : (uint32_t)start;
        int rc = replay_long_rows(c, &cap);
        if (rc) return rc;
        caps[k] = cap;
    }
    return ZDW_OK;
}

int zdw_shard_set_prev_row(zdw_ctx *c, const uint64_t *stored)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    c->have_prev = stored != nullptr;
    if (stored) c->prev_row.assign(stored, stored + c->ncols);
    if (c->p2_size_early && stored) c->p2_row0_stale = true;
    invalidate_pass2(c, true);
    return ZDW_OK;
}

int zdw_shard_last_row(zdw_ctx *c, uint64_t *stored)
{
    if (!c || !stored) return ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    int rc = ensure_dict(c);                    // stored values of text columns are dictionary offsets
    if (rc) return rc;
    if (c->nz_mode && !c->rec_dense && c->nrows) {      // the last row's records, whichever columns they are in: that row is completed from the map
        launch_rec_zero_fill(c->rec.as<uint8_t>() + (uint64_t)(c->nrows - 1) * c->rec_stride, c->rec_stride, 1, c->ncols, c->d_kind.as<uint8_t>(),
                             c->d_recoff.as<uint32_t>(), c->nz.as<uint32_t>() + (uint64_t)(c->nrows - 1) * c->nz_stride, c->nz_stride, c->stream);
    }
    Pass2Args a;
    rc = prepare_pass2(c, a);
    if (rc) return rc;
    // unused columns carry 0; a shard without rows passes its predecessor's row through
    for (uint32_t k = 0; k < c->ncols; ++k) stored[k] = (c->nrows == 0 && c->have_prev) ? c->prev_row[k] : 0;
    if (c->nrows && a.nused) {
        launch_last_row(a, c->stream);
        std::vector<uint64_t> last(a.nused);
        HIP_TRY(c, hipMemcpyAsync(last.data(), c->d_last.p, a.nused * 8ull, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        for (uint32_t u = 0; u < a.nused; ++u) stored[c->used[u]] = last[u];
    }
    return ZDW_OK;
}


// ---- range-partitioned dictionary: the node-wide dictionary is sorted once, 1/N of it on every rank -------
// Requester side of shard r:  sort own uniques -> sample 16-byte keys -> (splitters agreed by the caller) ->
// partition: strings in sorted order, part p = keys in [splitter[p-1], splitter[p]) -> all-to-all ->
// (owner side builds its range) -> offsets come back in the order the strings were sent -> set_offsets.
int zdw_shard_sort_uniques(zdw_ctx *c, uint64_t *count)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    P1Status st{};
    HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (st.id_next > c->id_cap) return fail(c, ZDW_UNKNOWN_ERROR, "unique id counter (%u) beyond its table (%u)", st.id_next, c->id_cap);
    const uint64_t u = st.id_next;
    c->presorted = experiment("ZDW_SHARD_PRESORT") != nullptr;      // the first form of this path (A/B measurements): full sort of the own uniques
    if (!c->presorted) {
        // The owners sort; a shard only has to know which range each of its strings falls in (zdw_shard_partition).
        if (st.entries > u) return fail(c, ZDW_UNKNOWN_ERROR, "more dictionary entries (%u) than unique ids (%u)", st.entries, st.id_next);
        HIP_TRY(c, c->my_order.ensure((u + 1) * 4));
        c->sorted_n = u;
        c->sorted_holes = u - st.entries;
        c->uniques_sorted = true;
        if (count) *count = st.entries;
        return ZDW_OK;
    }
    int rc = sort_strings(c, c->d_in, c->n, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), u,
                          std::max<uint32_t>(24, st.max_len), c->uniq_key.as<uint64_t>());
    if (rc) return rc;
    // keep the order: the owner-side sort reuses the sort buffers.  Holes (length 0) sort first.
    HIP_TRY(c, c->my_order.ensure((u + 1) * 4));
    HIP_TRY(c, hipMemcpyAsync(c->my_order.p, c->order.p, u * 4, hipMemcpyDeviceToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    c->sorted_n = u;
    c->sorted_holes = u - st.entries;
    {   // holes sort first and `entries` counts the strings: check the boundary rather than trust the arithmetic
        const uint64_t H = c->sorted_holes;
        uint32_t ids[2] = {0, 0}, lens[2] = {0, 1};
        if (st.entries > u) return fail(c, ZDW_UNKNOWN_ERROR, "more dictionary entries (%u) than unique ids (%u)", st.entries, st.id_next);
        if (H > 0) HIP_TRY(c, hipMemcpy(&ids[0], c->my_order.as<uint32_t>() + (H - 1), 4, hipMemcpyDeviceToHost));
        if (H < u) HIP_TRY(c, hipMemcpy(&ids[1], c->my_order.as<uint32_t>() + H, 4, hipMemcpyDeviceToHost));
        if (H > 0) HIP_TRY(c, hipMemcpy(&lens[0], c->uniq_len.as<uint32_t>() + ids[0], 4, hipMemcpyDeviceToHost));
        if (H < u) HIP_TRY(c, hipMemcpy(&lens[1], c->uniq_len.as<uint32_t>() + ids[1], 4, hipMemcpyDeviceToHost));
        if (lens[0] != 0 || lens[1] == 0) return fail(c, ZDW_UNKNOWN_ERROR, "unique-id holes (%llu) do not match the entry count", (unsigned long long)H);
    }
    c->uniques_sorted = true;
    if (count) *count = st.entries;
    return ZDW_OK;
}

int zdw_shard_sample_keys(zdw_ctx *c, uint32_t nsamples, uint64_t *keys)
{
    if (!c || !c->uniques_sorted) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_sort_uniques has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    const uint64_t real = c->sorted_n - c->sorted_holes;
    for (uint32_t i = 0; i < 2 * nsamples; ++i) keys[i] = ~0ull;           // a shard without strings samples "infinity"
    if (!real || !nsamples) return ZDW_OK;
    std::vector<uint32_t> idx(nsamples);
    HIP_TRY(c, c->part_idx.ensure(nsamples * 4)); HIP_TRY(c, c->part_keys.ensure(nsamples * 16));
    if (!c->presorted) {
        // unsorted uniques: ids at a fixed stride (ids follow first appearance in the shard, which says nothing about
        // a string's place in the order), each moved on to the next real string
        for (uint32_t i = 0; i < nsamples; ++i) idx[i] = (uint32_t)(((2ull * i + 1) * c->sorted_n) / (2ull * nsamples));
        HIP_TRY(c, hipMemcpyAsync(c->part_idx.p, idx.data(), nsamples * 4, hipMemcpyHostToDevice, s));
        launch_key16_sample(c->d_in, c->n, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), c->part_idx.as<uint32_t>(),
                            nsamples, (uint32_t)c->sorted_n, c->part_keys.as<uint64_t>(), s);
        HIP_TRY(c, hipMemcpyAsync(keys, c->part_keys.p, nsamples * 16, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        return ZDW_OK;
    }
    for (uint32_t i = 0; i < nsamples; ++i) idx[i] = (uint32_t)(c->sorted_holes + (uint64_t)(i + 1) * real / (nsamples + 1));
    for (auto &x : idx) if (x >= c->sorted_n) x = (uint32_t)c->sorted_n - 1;
    HIP_TRY(c, hipMemcpyAsync(c->part_idx.p, idx.data(), nsamples * 4, hipMemcpyHostToDevice, s));
    launch_key16_at(c->d_in, c->n, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), c->my_order.as<uint32_t>(),
                    c->part_idx.as<uint32_t>(), nsamples, c->part_keys.as<uint64_t>(), s);
    HIP_TRY(c, hipMemcpyAsync(keys, c->part_keys.p, nsamples * 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    return ZDW_OK;
}

int zdw_shard_partition(zdw_ctx *c, uint32_t nparts, const uint64_t *splitters, uint64_t *counts, uint64_t *nbytes, const void **d_lens,
                        const void **d_bytes)
{
    if (!c || !c->uniques_sorted) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_sort_uniques has not run") : ZDW_UNKNOWN_ERROR;
    if (!nparts || !counts || !nbytes) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_partition: bad arguments");
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    const uint64_t real = c->sorted_n - c->sorted_holes;
    std::vector<uint32_t> bounds(nparts + 1, 0);
    bounds[nparts] = (uint32_t)real;
    if (!c->presorted) {
        // group the ids by range (holes first, then part 0, 1, ...): one stable radix pass on the range number
        SortBuffers b{};
        int rcb = sort_buffers(c, c->sorted_n, b);
        if (rcb) return rcb;
        HIP_TRY(c, c->part_split.ensure(nparts * 16 + 16)); HIP_TRY(c, c->part_bounds.ensure((nparts + 2) * 4));
        if (nparts > 1) HIP_TRY(c, hipMemcpyAsync(c->part_split.p, splitters, (nparts - 1) * 16, hipMemcpyHostToDevice, s));
        launch_bucket_partition(c->d_in, c->n, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), (uint32_t)c->sorted_n,
                                c->part_split.as<uint64_t>(), nparts, b, c->my_order.as<uint32_t>(), c->part_bounds.as<uint32_t>(), s,
                                c->uniq_key.as<uint64_t>());
        std::vector<uint32_t> gb(nparts + 2, 0);
        HIP_TRY(c, hipMemcpyAsync(gb.data(), c->part_bounds.p, (nparts + 2) * 4, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (gb[1] != c->sorted_holes) return fail(c, ZDW_UNKNOWN_ERROR, "unique-id holes (%u) do not match the entry count", gb[1]);
        for (uint32_t p = 0; p < nparts; ++p) bounds[p] = gb[p + 1] - gb[1];
    }
    const uint32_t *order = c->my_order.as<uint32_t>() + c->sorted_holes;      // the real strings, grouped by range
    HIP_TRY(c, c->lenp1.ensure((real + 1) * 4)); HIP_TRY(c, c->off_excl.ensure((real + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(real + 1)));
    HIP_TRY(c, c->part_lens.ensure((real + 1) * 4));
    launch_dict_lengths(order, c->uniq_len.as<uint32_t>(), real, c->lenp1.as<uint32_t>(), s);
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), real, c->scan_tmp.p, nullptr, s);
    // part boundaries (sorted uniques): first sorted position whose 16-byte key reaches each splitter
    if (c->presorted && nparts > 1) {
        HIP_TRY(c, c->part_split.ensure((nparts - 1) * 16)); HIP_TRY(c, c->part_bounds.ensure((nparts - 1) * 4));
        HIP_TRY(c, hipMemcpyAsync(c->part_split.p, splitters, (nparts - 1) * 16, hipMemcpyHostToDevice, s));
        launch_partition_search(c->d_in, c->n, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), order, 0,
                                (uint32_t)real, c->part_split.as<uint64_t>(), nparts - 1, c->part_bounds.as<uint32_t>(), s);
        HIP_TRY(c, hipMemcpyAsync(bounds.data() + 1, c->part_bounds.p, (nparts - 1) * 4, hipMemcpyDeviceToHost, s));
    }
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->off_excl.as<uint64_t>() + real, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    for (uint32_t p = 1; p <= nparts; ++p) if (bounds[p] < bounds[p - 1]) bounds[p] = bounds[p - 1];      // splitters arrive sorted; be safe
    std::vector<uint64_t> offs(nparts + 1, 0);
    for (uint32_t p = 1; p < nparts; ++p)
        HIP_TRY(c, hipMemcpyAsync(&offs[p], c->off_excl.as<uint64_t>() + bounds[p], 8, hipMemcpyDeviceToHost, s));
    offs[nparts] = total;
    HIP_TRY(c, c->exp_bytes.ensure(total + 16));
    launch_export_strings(c->d_in, c->extra.as<uint8_t>(), c->uniq_pos.as<uint64_t>(), c->uniq_len.as<uint32_t>(), order, c->off_excl.as<uint64_t>(),
                          real, c->exp_bytes.as<uint8_t>(), c->part_lens.as<uint32_t>(), s, c->uniq_key.as<uint64_t>());
    HIP_TRY(c, hipStreamSynchronize(s));
    for (uint32_t p = 0; p < nparts; ++p) { counts[p] = bounds[p + 1] - bounds[p]; nbytes[p] = offs[p + 1] - offs[p]; }
    if (d_lens) *d_lens = c->part_lens.p;
    if (d_bytes) *d_bytes = c->exp_bytes.p;
    return ZDW_OK;
}

// Owner side: the strings of one key range from every shard (lens + NUL-terminated bytes, any order, duplicates
// across shards): dedupe (zdw_owner_import), then sort and lay out (zdw_owner_finish).  slice_bytes = the bytes this range takes in
// the dictionary.  With positions (where a memory limit may end the block, include/zdw_hip.h): the smallest position reported for a
// string is kept, the arrival that reported it is its first occurrence node-wide (zdw_owner_first_flags), and strings first met
// beyond the block's end are dropped before the sort (zdw_owner_finish's last_pos).
int zdw_owner_import(zdw_ctx *c, const void *d_lens, const void *d_bytes, const void *d_pos, uint64_t count, uint64_t nbytes)
{
    if (!c) return ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    c->own_built = c->own_imported = false; c->own_count = count; c->own_slice = 0; c->own_nids = 0; c->own_has_pos = false;
    if (nbytes >= EXTRA_BASE || count >= 0xFFFFFFF0ull) return fail(c, ZDW_OUT_OF_MEMORY, "dictionary range too large");
    uint64_t nslots = 1ull << 14;
    while (nslots < 2 * count) nslots <<= 1;
    const uint64_t cap = count + 64ull * 16384 + 64;
    HIP_TRY(c, c->own_slots.ensure(nslots * 32)); HIP_TRY(c, c->own_upos.ensure(cap * 8)); HIP_TRY(c, c->own_ulen.ensure(cap * 4));
    if (c->budget.cap || c->own_key.ensure(cap * 24) != hipSuccess) c->own_key.release();      // (optional: ensure_dict's note on uniq_key)
    HIP_TRY(c, c->own_status.ensure(sizeof(P1Status)));
    HIP_TRY(c, c->own_extra.ensure(nbytes + 16)); HIP_TRY(c, c->own_lens.ensure((count + 1) * 4)); HIP_TRY(c, c->own_offs.ensure((count + 2) * 8));
    HIP_TRY(c, c->own_ids.ensure((count + 1) * 4));
    HIP_TRY(c, c->lenp1.ensure((cap + 1) * 4)); HIP_TRY(c, c->off_excl.ensure((cap + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(cap + 1)));
    HIP_TRY(c, hipMemsetAsync(c->own_slots.p, 0, nslots * 32, s));
    P1Status init{};
    init.bad_row = 0xFFFFFFFFu;
    HIP_TRY(c, hipMemcpyAsync(c->own_status.p, &init, sizeof init, hipMemcpyHostToDevice, s));
    if (count) {
        HIP_TRY(c, hipMemcpyAsync(c->own_extra.p, d_bytes, nbytes, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->own_lens.p, d_lens, count * 4, hipMemcpyDeviceToDevice, s));
    }
    launch_dict_lengths(nullptr, c->own_lens.as<uint32_t>(), count, c->lenp1.as<uint32_t>(), s);
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->own_offs.as<uint64_t>(), count, c->scan_tmp.p, nullptr, s);
    uint64_t total_in = 0;
    HIP_TRY(c, hipMemcpyAsync(&total_in, c->own_offs.as<uint64_t>() + count, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (total_in != nbytes)
        return fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_import: lengths add up to %llu bytes, %llu were passed", (unsigned long long)total_in,
                    (unsigned long long)nbytes);
    launch_import_strings(nullptr, 0, c->own_extra.as<uint8_t>(), c->own_lens.as<uint32_t>(), c->own_offs.as<uint64_t>(), count,
                          c->own_slots.as<uint64_t>(), nslots - 1, (uint32_t)std::min<uint64_t>(nslots - nslots / 8, 0xFFFFFFF0ull),
                          c->own_upos.as<uint64_t>(), c->own_ulen.as<uint32_t>(), (uint32_t)cap, 0u, 0, c->own_status.as<P1Status>(),
                          c->own_ids.as<uint32_t>(), s, c->own_key.as<uint64_t>());
    P1Status st{};
    HIP_TRY(c, hipMemcpyAsync(&st, c->own_status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (st.overflow || st.id_next > cap) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_import: range table overflow");
    c->own_nids = st.id_next;
    c->own_max_len = std::max<uint32_t>(24, st.max_len);
    if (d_pos) {
        HIP_TRY(c, c->own_pos.ensure((count + 1) * 8)); HIP_TRY(c, c->own_minpos.ensure((c->own_nids + 1) * 8));
        if (count) HIP_TRY(c, hipMemcpyAsync(c->own_pos.p, d_pos, count * 8, hipMemcpyDeviceToDevice, s));
        HIP_TRY(c, hipMemsetAsync(c->own_minpos.p, 0xFF, (c->own_nids + 1) * 8, s));
        launch_owner_minpos(c->own_ids.as<uint32_t>(), c->own_pos.as<uint64_t>(), count, c->own_minpos.as<uint64_t>(), s);
        HIP_TRY(c, hipStreamSynchronize(s));
        c->own_has_pos = true;
    }
    c->own_imported = true;
    return ZDW_OK;
}

// per arrival (u32): 1 = the arrival's position is the smallest any shard reported for its string
int zdw_owner_first_flags(zdw_ctx *c, const void **d_flags)
{
    if (!c || !c->own_imported || !c->own_has_pos) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_import has not run with positions") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    HIP_TRY(c, c->own_flags.ensure((c->own_count + 1) * 4));
    launch_owner_first(c->own_ids.as<uint32_t>(), c->own_pos.as<uint64_t>(), c->own_count, c->own_minpos.as<uint64_t>(), c->own_flags.as<uint32_t>(), c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d_flags) *d_flags = c->own_flags.p;
    return ZDW_OK;
}

int zdw_owner_finish(zdw_ctx *c, uint64_t last_pos, uint64_t *slice_bytes)
{
    if (!c || !c->own_imported) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_import has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    const uint64_t u = c->own_nids;
    if (last_pos != ~0ull) {
        if (!c->own_has_pos) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_finish: the strings came without positions");
        launch_owner_filter(c->own_minpos.as<uint64_t>(), c->own_ulen.as<uint32_t>(), (uint32_t)u, last_pos, s);
    }
    int rc = sort_strings(c, nullptr, 0, c->own_extra.as<uint8_t>(), c->own_upos.as<uint64_t>(), c->own_ulen.as<uint32_t>(), u, c->own_max_len,
                          c->own_key.as<uint64_t>());
    if (rc) return rc;
    {
        StageScope ts(c, ZDW_STAGE_DICT_EMIT);
        launch_dict_lengths(c->order.as<uint32_t>(), c->own_ulen.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);
        launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
        uint64_t total = 0;
        HIP_TRY(c, hipMemcpyAsync(&total, c->off_excl.as<uint64_t>() + u, 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        if (total >= (1ull << 32)) return fail(c, ZDW_OUT_OF_MEMORY, "dictionary range exceeds 2^32 bytes: split the block");
        HIP_TRY(c, c->own_blob.ensure(16 + total + 16));
        HIP_TRY(c, c->own_idoff.ensure((c->own_nids + 1) * 4));
        // (strings zdw_owner_finish dropped -- first met beyond the block's end -- are holes the layout skips: their entries read 0, not
        // whatever the buffer held, so zdw_owner_offsets answers "no offset" for them: a deterministic value a test can look for)
        HIP_TRY(c, hipMemsetAsync(c->own_idoff.p, 0, (c->own_nids + 1) * 4, s));
        // blob[0] is the byte before the range (the origin byte when this is the first range); local offsets start at 1
        launch_dict_emit(nullptr, 0, c->own_extra.as<uint8_t>(), c->order.as<uint32_t>(), c->own_upos.as<uint64_t>(), c->own_ulen.as<uint32_t>(),
                         c->off_excl.as<uint64_t>(), u, c->own_blob.as<uint8_t>() + 8, c->own_idoff.as<uint32_t>(), s, c->own_key.as<uint64_t>());
        HIP_TRY(c, hipStreamSynchronize(s));
        c->own_slice = total;
    }
    c->own_built = true;
    if (slice_bytes) *slice_bytes = c->own_slice;
    return ZDW_OK;
}

int zdw_owner_build(zdw_ctx *c, const void *d_lens, const void *d_bytes, uint64_t count, uint64_t nbytes, uint64_t *slice_bytes)
{
    const int rc = zdw_owner_import(c, d_lens, d_bytes, nullptr, count, nbytes);
    return rc ? rc : zdw_owner_finish(c, ~0ull, slice_bytes);
}

// dictionary offsets of the strings zdw_owner_build received, in arrival order; base = offset of the range's first string
int zdw_owner_offsets(zdw_ctx *c, uint64_t base, const void **d_offsets)
{
    if (!c || !c->own_built) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_build has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    if (base == 0 || base + c->own_slice > (1ull << 32)) return fail(c, ZDW_OUT_OF_MEMORY, "dictionary exceeds 2^32 bytes: split the block");
    HIP_TRY(c, c->own_out.ensure((c->own_count + 1) * 4));
    launch_owner_offsets(c->own_ids.as<uint32_t>(), c->own_idoff.as<uint32_t>(), (uint32_t)(base - 1), c->own_count, c->own_out.as<uint32_t>(),
                         c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d_offsets) *d_offsets = c->own_out.p;
    return ZDW_OK;
}

// Requester side, last step: offsets of this shard's strings (in the order zdw_shard_partition sent them);
// dict_size = size of the whole dictionary.  The shard's dictionary section becomes the range it owns
// (with the section header when it owns the first range): sections of ranks 0..N-1 concatenate to the file's.
int zdw_shard_set_offsets(zdw_ctx *c, const void *d_offsets, uint64_t count, uint64_t dict_size, int first_range)
{
    if (!c || !c->uniques_sorted || !c->own_built) return c ? fail(c, ZDW_UNKNOWN_ERROR, "partitioned dictionary steps out of order") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    const uint64_t real = c->sorted_n - c->sorted_holes;
    if (count != real) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_set_offsets: %llu offsets for %llu strings", (unsigned long long)count,
                                   (unsigned long long)real);
    const uint64_t limit = c->max_dict_bytes && c->max_dict_bytes < (1ull << 32) ? c->max_dict_bytes : (1ull << 32) - 1;
    if (dict_size > limit)
        return fail(c, ZDW_OUT_OF_MEMORY, "dictionary of %llu bytes exceeds the limit of %llu: split the block (smaller zdw_split_hint.max_rows)",
                    (unsigned long long)dict_size, (unsigned long long)limit);
    HIP_TRY(c, c->slot_off.ensure((c->sorted_n + 1) * 4));
    launch_scatter_offsets(c->my_order.as<uint32_t>() + c->sorted_holes, static_cast<const uint32_t *>(d_offsets), real, c->slot_off.as<uint32_t>(), s);
    c->nuniq = dict_size > 1 ? std::max<uint64_t>(real, 1) : 0;
    return zdw_owner_section(c, dict_size, first_range);
}

int zdw_owner_section(zdw_ctx *c, uint64_t dict_size, int first_range)
{
    if (!c || !c->own_built) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_finish has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    c->dict_size = dict_size;
    c->index_size = (uint8_t)bytes_for(dict_size);
    // this rank's piece of the dictionary section
    HIP_TRY(c, c->dict_sec.ensure(16 + c->own_slice + 16));
    uint8_t *sec = c->dict_sec.as<uint8_t>();
    HIP_TRY(c, hipMemcpyAsync(sec + 8, c->own_blob.as<uint8_t>() + 8, 1 + c->own_slice, hipMemcpyDeviceToDevice, s));
    if (first_range) {
        if (dict_size <= 1) {                       // empty set: a single 00 (dictionary.cpp:80-84)
            c->dict_sec_off = 8; c->dict_sec_len = 1;
        } else {
            uint8_t hdr[8];
            hdr[0] = c->index_size;
            for (unsigned i = 0; i < c->index_size; ++i) hdr[1 + i] = (uint8_t)(dict_size >> (8 * i));
            c->dict_sec_off = 8 - 1 - c->index_size;
            c->dict_sec_len = 1 + c->index_size + 1 + c->own_slice;
            HIP_TRY(c, hipMemcpyAsync(sec + c->dict_sec_off, hdr, 1 + c->index_size, hipMemcpyHostToDevice, s));
        }
    } else {
        c->dict_sec_off = 9; c->dict_sec_len = c->own_slice;
    }
    HIP_TRY(c, hipStreamSynchronize(s));
    if (!c->uniques_sorted) c->nuniq = dict_size > 1 ? 1 : 0;          // (no rows of this block here: only "is the section empty" is ever asked)
    c->dict_size_done = c->dict_done = true;
    c->stats_done = false;
    invalidate_pass2(c, c->p2_size_early && c->p2_early_size == dict_size);      // (row lengths started early for this very size stay)
    return ZDW_OK;
}

// (owner) the bytes this range will take in the dictionary -- the sum of len + 1 over its strings -- known once the arrivals are
// deduplicated, BEFORE they are sorted: with every owner's figure the requesters know the dictionary's size (the width of the text
// columns) and start their row lengths while the owners sort (zdw_shard_rows_early)
int zdw_owner_size(zdw_ctx *c, uint64_t *bytes)
{
    if (!c || !c->own_imported) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_owner_import has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    uint64_t total = 0;
    if (c->own_nids) {
        HIP_TRY(c, c->scan_tmp.ensure(std::max<size_t>(scan_tmp_bytes(c->own_nids + 1), DICT_TOTAL_BLOCKS * 8)));
        const uint32_t nb = launch_dict_total(c->own_ulen.as<uint32_t>(), c->own_nids, static_cast<uint64_t *>(c->scan_tmp.p), s);
        std::vector<uint64_t> part(nb);
        HIP_TRY(c, hipMemcpyAsync(part.data(), c->scan_tmp.p, (size_t)nb * 8, hipMemcpyDeviceToHost, s));
        HIP_TRY(c, hipStreamSynchronize(s));
        for (uint64_t v : part) total += v;
    }
    if (bytes) *bytes = total;
    return ZDW_OK;
}

// (requester) the block's dictionary will be dict_size bytes (origin byte included): the row lengths of this shard's rows start now, on
// the context's second stream, as if no row preceded its first one; zdw_shard_set_prev_row later has that one row done again.  The
// column ranges must be the block's (zdw_shard_import_ranges) already.
int zdw_shard_rows_early(zdw_ctx *c, uint64_t dict_size)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    if (c->have_prev || c->p2_size_done || c->p2_done || dict_size == 0 || dict_size >= (1ull << 32)) return ZDW_OK;      // (nothing to gain, or not representable: the ordinary way later)
    c->dict_size = dict_size;
    c->index_size = (uint8_t)bytes_for(dict_size);
    c->dict_size_done = true;
    c->stats_done = false;
    const int rc = start_pass2_size(c);
    if (rc) return rc;
    if (c->p2_size_done) { c->p2_size_early = true; c->p2_early_size = dict_size; }
    return ZDW_OK;
}

// ---- where the reference's memory limit ends a row-sharded block (include/zdw_hip.h) ----
double zdw_model_upper_bound_mb(uint64_t entries, uint64_t bytes, uint32_t max_len)
{
    // even with every arena wasting its tail (a string never straddles two arenas: stringheap.cpp:61-73)
    const uint64_t ml = (uint64_t)std::max<uint32_t>(max_len, 24) + 1;
    const double arenas = (double)(bytes / (MODEL_ARENA > 2 * ml ? MODEL_ARENA - ml : MODEL_ARENA / 2) + 1);
    return MODEL_BASE_MB + 64.0 * arenas + MODEL_NODE_BYTES * (double)entries / 1048576.0;
}

int zdw_shard_model_counts(zdw_ctx *c, uint64_t *entries, uint64_t *bytes, uint32_t *max_len)
{
    if (!c || !c->p1_done) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    P1Status st{};
    HIP_TRY(c, hipMemcpyAsync(&st, c->status.p, sizeof st, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    const uint64_t u = st.id_next;
    if (u > c->id_cap) return fail(c, ZDW_UNKNOWN_ERROR, "unique id counter (%u) beyond its table (%u)", st.id_next, c->id_cap);
    HIP_TRY(c, c->lenp1.ensure((u + 1) * 4)); HIP_TRY(c, c->off_excl.ensure((u + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(u + 1)));
    launch_dict_lengths(nullptr, c->uniq_len.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
    uint64_t total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, c->off_excl.as<uint64_t>() + u, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    if (entries) *entries = st.entries;
    if (bytes) *bytes = total;
    if (max_len) *max_len = st.max_len;
    return ZDW_OK;
}

int zdw_shard_positions(zdw_ctx *c, uint64_t base, const void **d_pos)
{
    if (!c || !c->uniques_sorted) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_partition has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    const uint64_t real = c->sorted_n - c->sorted_holes;
    HIP_TRY(c, c->exp_pos.ensure((real + 1) * 8));
    launch_export_positions(c->my_order.as<uint32_t>() + c->sorted_holes, c->uniq_pos.as<uint64_t>(), (uint32_t)real, base, c->exp_pos.as<uint64_t>(), c->stream);
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (d_pos) *d_pos = c->exp_pos.p;
    return ZDW_OK;
}

int zdw_shard_model_walk(zdw_ctx *c, const void *d_flags, uint64_t count, double limit_mb, zdw_model_state *state, uint32_t *found, uint32_t *row,
                         uint32_t *fields, uint64_t *row_start, uint64_t *trip_pos)
{
    if (!c || !c->uniques_sorted || !state || !found) return c ? fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_partition has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    hipStream_t s = c->stream;
    *found = 0;
    const uint64_t real = c->sorted_n - c->sorted_holes, u = c->sorted_n;
    if (count != real) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_model_walk: %llu flags for %llu strings", (unsigned long long)count, (unsigned long long)real);
    if (!real) return ZDW_OK;
    // this shard's node-wide-first strings by position, prefix sums of len + 1 in that order, the walk over the arenas
    HIP_TRY(c, c->model_len.ensure((u + 1) * 4));
    HIP_TRY(c, c->d_scalar.ensure(64));
    HIP_TRY(c, hipMemsetAsync(c->model_len.p, 0, (u + 1) * 4, s));
    HIP_TRY(c, hipMemsetAsync(c->d_scalar.p, 0, 64, s));
    launch_mask_first(c->my_order.as<uint32_t>() + c->sorted_holes, static_cast<const uint32_t *>(d_flags), (uint32_t)real, c->uniq_len.as<uint32_t>(),
                      c->model_len.as<uint32_t>(), c->d_scalar.as<uint32_t>() + 12, s);
    uint32_t nfirst = 0;
    HIP_TRY(c, hipMemcpyAsync(&nfirst, c->d_scalar.as<uint32_t>() + 12, 4, hipMemcpyDeviceToHost, s));
    SortBuffers b{};
    int rc = sort_buffers(c, u, b);
    if (rc) return rc;
    HIP_TRY(c, c->lenp1.ensure((u + 1) * 4)); HIP_TRY(c, c->off_excl.ensure((u + 2) * 8));
    HIP_TRY(c, c->scan_tmp.ensure(scan_tmp_bytes(u + 1)));
    uint32_t *by_pos = nullptr;
    launch_model_sort(c->uniq_pos.as<uint64_t>(), c->model_len.as<uint32_t>(), (uint32_t)u, b, &by_pos, s);
    launch_dict_lengths(by_pos, c->model_len.as<uint32_t>(), u, c->lenp1.as<uint32_t>(), s);
    launch_scan_u32_to_u64(c->lenp1.as<uint32_t>(), c->off_excl.as<uint64_t>(), u, c->scan_tmp.p, nullptr, s);
    HIP_TRY(c, hipStreamSynchronize(s));          // nfirst
    launch_model_trip_chained(c->off_excl.as<uint64_t>(), nfirst, state->arenas, state->entries, state->arena_free, MODEL_BASE_MB, MODEL_NODE_BYTES, limit_mb,
                              MODEL_ARENA, c->d_scalar.as<uint64_t>(), s);
    uint64_t res[4] = {~0ull, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(res, c->d_scalar.p, 32, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    state->arenas = res[1]; state->entries = res[2]; state->arena_free = res[3];
    if (res[0] == ~0ull) return ZDW_OK;
    uint32_t id = 0;
    uint64_t pos = 0;
    HIP_TRY(c, hipMemcpy(&id, by_pos + res[0], 4, hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(&pos, c->uniq_pos.as<uint64_t>() + id, 8, hipMemcpyDeviceToHost));
    pos &= POS_MASK;
    if (pos >= EXTRA_BASE) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_model_walk: the tripping string has no position in this shard's image");
    uint32_t r = 0, f = 0;
    uint64_t rs = 0;
    rc = locate_occurrence(c, pos, c->last_span_shift, c->last_span0, &r, &f, &rs);
    if (rc) return rc;
    *found = 1;
    if (row) *row = r;
    if (fields) *fields = f;
    if (row_start) *row_start = rs;
    if (trip_pos) *trip_pos = pos;
    return ZDW_OK;
}

int zdw_shard_trim(zdw_ctx *c, uint32_t row, uint32_t fields, uint64_t row_start, zdw_pass1_result *res)
{
    if (!c || !c->p1_done || !res) return c ? fail(c, ZDW_UNKNOWN_ERROR, "pass 1 has not run") : ZDW_UNKNOWN_ERROR;
    (void)hipSetDevice(c->device);
    if (row >= c->nrows) return fail(c, ZDW_UNKNOWN_ERROR, "zdw_shard_trim: row %u is not among the shard's %u rows", row, c->nrows);
    // (a shard that keeps no row still keeps the partial row's fields in the block's dictionary and ranges: no error here, the
    // caller knows whether the whole block is left without a row)
    const int rc = trim_block(c, row, fields, row_start, res, true);
    if (rc) return rc;
    c->dict_size_done = c->dict_done = c->stats_done = false;
    invalidate_pass2(c);
    return ZDW_OK;
}

}  // extern "C"
