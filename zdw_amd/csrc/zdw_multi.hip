// zdw_multi.hip — one ZDW block row-sharded over several GPUs of one node, inside the library (include/zdw_hip.h,
// zdw_multi_*): what convertDWfile --gpus=N runs.  SURVEY.md §8e; the reference has no counterpart (one process, one core).
//
// One process, one zdw_ctx and one host thread per device.  Every shard holds a contiguous row range of ONE block whose
// bytes must not depend on the number of shards, so after each shard's pass 1:
//   1. column ranges        min / max / seen  -> ncclAllReduce (min, max, max) on every device
//   2. dictionary           range-partitioned: every shard samples 16-byte prefix keys of its unique strings, the host
//                           picks N - 1 splitters, strings travel to the owner of their key range (grouped ncclSend /
//                           ncclRecv over the xGMI links: an all-to-all-v), the owner dedupes + sorts + lays out its range,
//                           every string's offset travels back the same way.  The node-wide dictionary is sorted once,
//                           1/N of it per device, and device r's section piece is range r
//   3. row halo             stored values of a shard's last row -> "previous row" of the next shard (a few KB: host)
// Sizes, counts, splitters and status codes are plain host values: all shards live in this process.
//
// Transport: RCCL (ncclCommInitAll, grouped point-to-point) when the devices are distinct; when the same device is
// listed more than once (tests on a one-GPU box) the same exchange runs as device-to-device copies.
#include "../../include/zdw_hip.h"
#include "zdw_knobs.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the library is loaded on first use (below)
#include <dlfcn.h>

#include <algorithm>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <string>
#include <thread>
#include <vector>

namespace {

// RCCL is taken from the process when it is already there (a host application -- PyTorch, say -- brings its own copy, and two
// copies of it in one process end in a double free at exit), else loaded from the ROCm installation; a context that never
// talks to a second device never loads it.
struct Rccl {
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    bool ok = false;
};
const Rccl &rccl()
{
    static Rccl r = [] {
        Rccl x;
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
        if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!h) return x;
        x.CommInitAll = reinterpret_cast<decltype(x.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
        x.CommDestroy = reinterpret_cast<decltype(x.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
        x.GetErrorString = reinterpret_cast<decltype(x.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
        x.GroupStart = reinterpret_cast<decltype(x.GroupStart)>(dlsym(h, "ncclGroupStart"));
        x.GroupEnd = reinterpret_cast<decltype(x.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
        x.AllReduce = reinterpret_cast<decltype(x.AllReduce)>(dlsym(h, "ncclAllReduce"));
        x.Send = reinterpret_cast<decltype(x.Send)>(dlsym(h, "ncclSend"));
        x.Recv = reinterpret_cast<decltype(x.Recv)>(dlsym(h, "ncclRecv"));
        x.ok = x.CommInitAll && x.CommDestroy && x.GetErrorString && x.GroupStart && x.GroupEnd && x.AllReduce && x.Send && x.Recv;
        return x;
    }();
    return r;
}

struct Shard {
    zdw_ctx *ctx = nullptr;
    int device = 0;
    hipStream_t stream = nullptr;          // the exchange's stream on this device
    ncclComm_t comm = nullptr;
    uint64_t in_off = 0, in_len = 0;       // this shard's piece of the host image
    zdw_pass1_result p1{};
    int rc = 0;
    std::string err;
    // exchange buffers (device)
    void *d_ranges = nullptr;              // 3 x ncols u64: ~min, max, seen
    void *d_recv_lens = nullptr, *d_recv_bytes = nullptr, *d_recv_offs = nullptr, *d_back = nullptr, *d_recv_pos = nullptr;
    size_t cap_recv_lens = 0, cap_recv_bytes = 0, cap_recv_offs = 0, cap_back = 0, cap_recv_pos = 0;
    bool p1_valid = false;                 // pass 1 has run over the rows this shard still holds and nothing has cut them back since
    bool mc_valid = false;                 // ... and its share of the memory model's upper bound (zdw_shard_model_counts: a scan and two waits) is known
    uint64_t mc_entries = 0, mc_bytes = 0;
    uint32_t mc_max_len = 0;
    // partition results
    std::vector<uint64_t> counts, nbytes;  // per destination range
    const void *d_lens = nullptr, *d_bytes = nullptr, *d_pos = nullptr;
    const void *d_offsets = nullptr;
    uint64_t slice = 0;
};

}  // namespace

struct zdw_multi {
    std::vector<Shard> sh;
    uint32_t ncols = 0;
    bool use_rccl = false;
    std::string err;
    const uint8_t *image = nullptr;
    uint64_t image_len = 0;
    bool encoded = false;                  // a block has been encoded and not been left yet (zdw_multi_next)
    bool finished = false;                 // the input's last block has been left
    uint32_t longest_line = 16384;         // m_LongestLine: persists across the blocks of one input
    double mem_limit_mb = 0.0;             // zdw_opts.mem_limit_mb: where the reference's blocks end (0 = one block)
    size_t first = 0;                      // the first shard that still holds rows
    size_t blk_first = 0, blk_end = 0;     // the shards that hold the current block's rows
    bool blk_tripped = false;              // ... which the memory limit ended
};

namespace {

thread_local std::string g_multi_create_error;

int mfail(zdw_multi *m, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (m) m->err = buf; else g_multi_create_error = buf;
    return code;
}

#define M_HIP(m, call)                                                                                                   \
    do {                                                                                                                 \
        hipError_t e_ = (call);                                                                                          \
        if (e_ != hipSuccess) return mfail((m), e_ == hipErrorOutOfMemory ? ZDW_OUT_OF_MEMORY : ZDW_UNKNOWN_ERROR, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
#define M_NCCL(m, call)                                                                                                  \
    do {                                                                                                                 \
        ncclResult_t r_ = (call);                                                                                        \
        if (r_ != ncclSuccess) return mfail((m), ZDW_UNKNOWN_ERROR, "%s: %s", #call, rccl().GetErrorString(r_));        \
    } while (0)

int grow(zdw_multi *m, int device, void **p, size_t *cap, size_t bytes)
{
    if (bytes <= *cap && *p) return ZDW_OK;
    M_HIP(m, hipSetDevice(device));
    if (*p) (void)hipFree(*p);
    *p = nullptr; *cap = 0;
    const size_t want = bytes < 256 ? 256 : bytes + bytes / 16;
    M_HIP(m, hipMalloc(p, want));
    *cap = want;
    return ZDW_OK;
}

// run f(shard index) on every shard, one host thread per device (the per-shard calls block on their device)
template <typename F>
void for_shards(zdw_multi *m, F f)
{
    if (m->sh.size() == 1) { f(0); return; }
    std::vector<std::thread> th;
    for (size_t i = 0; i < m->sh.size(); ++i) th.emplace_back([&, i] { (void)hipSetDevice(m->sh[i].device); f(i); });
    for (auto &t : th) t.join();
}

int first_failure(zdw_multi *m, const char *what)
{
    for (size_t i = 0; i < m->sh.size(); ++i)
        if (m->sh[i].rc) return mfail(m, m->sh[i].rc, "%s failed on shard %zu (device %d): %s", what, i, m->sh[i].device, m->sh[i].err.c_str());
    return ZDW_OK;
}

// is byte p of the host image an unescaped newline (a row terminator)?  (getnextrow.cpp:45-53)
bool host_row_terminator(const uint8_t *in, uint64_t p)
{
    if (in[p] != '\n') return false;
    uint64_t k = 0;
    while (k < p && in[p - 1 - k] == '\\') ++k;
    return (k & 1) == 0;
}

// all-to-all-v between the shards: piece (q -> r) = cnt[q][r] elements of `esize` bytes, taken from src[q] at the running
// offset of shard q's pieces and delivered to dst[r] at the running offset of what r receives (pieces ordered by sender)
int all_to_all_v(zdw_multi *m, const std::vector<const void *> &src, const std::vector<void *> &dst, const std::vector<std::vector<uint64_t>> &cnt,
                 size_t esize)
{
    const size_t n = m->sh.size();
    std::vector<uint64_t> recv_at(n, 0);
    if (m->use_rccl) M_NCCL(m, rccl().GroupStart());
    for (size_t q = 0; q < n; ++q) {
        uint64_t send_at = 0;
        for (size_t r = 0; r < n; ++r) {
            const uint64_t c = cnt[q][r];
            if (c) {
                const char *from = static_cast<const char *>(src[q]) + send_at * esize;
                char *to = static_cast<char *>(dst[r]) + 0;
                // where this piece lands in r's buffer: after the pieces of senders 0 .. q - 1
                uint64_t land = 0;
                for (size_t p = 0; p < q; ++p) land += cnt[p][r];
                to += land * esize;
                if (m->use_rccl) {
                    M_NCCL(m, rccl().Send(from, c * esize, ncclUint8, (int)r, m->sh[q].comm, m->sh[q].stream));
                    M_NCCL(m, rccl().Recv(to, c * esize, ncclUint8, (int)q, m->sh[r].comm, m->sh[r].stream));
                } else {
                    M_HIP(m, hipMemcpyPeerAsync(to, m->sh[r].device, from, m->sh[q].device, c * esize, m->sh[r].stream));
                }
            }
            send_at += c;
        }
    }
    if (m->use_rccl) M_NCCL(m, rccl().GroupEnd());
    for (size_t r = 0; r < n; ++r) { M_HIP(m, hipSetDevice(m->sh[r].device)); M_HIP(m, hipStreamSynchronize(m->sh[r].stream)); }
    (void)recv_at;
    return ZDW_OK;
}

}  // namespace

extern "C" {

const char *zdw_multi_last_error(const zdw_multi *m) { return m ? m->err.c_str() : g_multi_create_error.c_str(); }

void zdw_multi_destroy(zdw_multi *m)
{
    if (!m) return;
    for (Shard &s : m->sh) {
        (void)hipSetDevice(s.device);
        if (s.ctx) zdw_ctx_destroy(s.ctx);
        for (void *p : {s.d_ranges, s.d_recv_lens, s.d_recv_bytes, s.d_recv_offs, s.d_back, s.d_recv_pos}) if (p) (void)hipFree(p);
        if (s.comm) (void)rccl().CommDestroy(s.comm);
        if (s.stream) (void)hipStreamDestroy(s.stream);
    }
    delete m;
}

int zdw_multi_create(const uint8_t *col_types, uint32_t ncols, const zdw_opts *opts, const int32_t *devices, uint32_t ndev, zdw_multi **out)
{
    if (!out || !devices || !ndev) return mfail(nullptr, ZDW_UNKNOWN_ERROR, "zdw_multi_create: bad arguments");
    *out = nullptr;
    zdw_multi *m = new (std::nothrow) zdw_multi();
    if (!m) return mfail(nullptr, ZDW_OUT_OF_MEMORY, "zdw_multi_create: out of host memory");
    m->ncols = ncols;
    m->sh.resize(ndev);
    std::set<int> distinct(devices, devices + ndev);
    const char *tr = zdw::experiment("ZDW_MULTI_TRANSPORT");      // "copy": device-to-device copies even between distinct devices (A/B measurements)
    m->use_rccl = distinct.size() == ndev && !(tr && !strcmp(tr, "copy"));
    for (uint32_t i = 0; i < ndev; ++i) {
        Shard &s = m->sh[i];
        s.device = devices[i];
        zdw_opts o;
        memset(&o, 0, sizeof o);
        if (opts) memcpy(&o, opts, std::min<size_t>(sizeof o, opts->struct_size ? opts->struct_size : sizeof o));
        o.struct_size = sizeof o;
        o.device = s.device;
        if (i == 0 && o.struct_size >= offsetof(zdw_opts, mem_limit_mb) + sizeof(uint64_t)) m->mem_limit_mb = (double)o.mem_limit_mb;
        o.mem_limit_mb = 0;                       // where a block ends is not a per-shard decision: zdw_multi_encode
        o.track_first_occurrence = m->mem_limit_mb > 0.0 ? 1 : 0;
        o.defer_bad_row = 1;                      // ... nor is whether the block reaches a bad row
        const int rc = zdw_ctx_create(col_types, ncols, &o, &s.ctx);
        if (rc) { const int r2 = mfail(nullptr, rc, "shard %u (device %d): %s", i, s.device, zdw_last_error(nullptr)); zdw_multi_destroy(m); return r2; }
        if (hipSetDevice(s.device) != hipSuccess || hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) {
            const int r2 = mfail(nullptr, ZDW_UNKNOWN_ERROR, "shard %u: no stream on device %d", i, s.device);
            zdw_multi_destroy(m);
            return r2;
        }
    }
    if (m->use_rccl) {
        if (!rccl().ok) { const int r2 = mfail(nullptr, ZDW_UNKNOWN_ERROR, "RCCL (librccl.so.1) is not available: %s", dlerror() ? dlerror() : "symbols missing"); zdw_multi_destroy(m); return r2; }
        std::vector<ncclComm_t> comms(ndev);
        std::vector<int> devs(devices, devices + ndev);
        const ncclResult_t r = rccl().CommInitAll(comms.data(), (int)ndev, devs.data());
        if (r != ncclSuccess) { const int r2 = mfail(nullptr, ZDW_UNKNOWN_ERROR, "ncclCommInitAll: %s", rccl().GetErrorString(r)); zdw_multi_destroy(m); return r2; }
        for (uint32_t i = 0; i < ndev; ++i) m->sh[i].comm = comms[i];
    }
    *out = m;
    return ZDW_OK;
}

// The host image is cut into one piece per shard at row boundaries (the first unescaped newline at or after i/N of it);
// every piece goes to its device.  The image must stay valid until zdw_multi_encode has returned.
int zdw_multi_set_input_host(zdw_multi *m, const uint8_t *tsv, uint64_t len)
{
    if (!m || (len && !tsv)) return m ? mfail(m, ZDW_UNKNOWN_ERROR, "zdw_multi_set_input_host: tsv is NULL") : ZDW_UNKNOWN_ERROR;
    const size_t n = m->sh.size();
    m->image = tsv; m->image_len = len; m->encoded = false; m->finished = false; m->first = 0; m->longest_line = 16384;
    for (Shard &s : m->sh) s.p1_valid = false;
    std::vector<uint64_t> cut(n + 1, len);
    cut[0] = 0;
    for (size_t i = 1; i < n; ++i) {
        uint64_t p = std::max<uint64_t>(len / n * i, cut[i - 1]);
        while (p < len && !host_row_terminator(tsv, p)) ++p;
        cut[i] = p < len ? p + 1 : len;
    }
    for (size_t i = 0; i < n; ++i) { m->sh[i].in_off = cut[i]; m->sh[i].in_len = cut[i + 1] - cut[i]; }
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        s.rc = zdw_set_input_host(s.ctx, tsv + s.in_off, s.in_len);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    return first_failure(m, "loading the input");
}

// The same for pieces that already lie on the devices (piece i on device i, 16-byte aligned, whole rows, in row order): nothing is
// copied, the contexts borrow the buffers until the next input.
int zdw_multi_set_input_device(zdw_multi *m, const void *const *d_pieces, const uint64_t *lens)
{
    if (!m || !d_pieces || !lens) return m ? mfail(m, ZDW_UNKNOWN_ERROR, "zdw_multi_set_input_device: bad arguments") : ZDW_UNKNOWN_ERROR;
    const size_t n = m->sh.size();
    m->image = nullptr; m->image_len = 0; m->encoded = false; m->finished = false; m->first = 0; m->longest_line = 16384;
    uint64_t at = 0;
    for (size_t i = 0; i < n; ++i) {
        Shard &s = m->sh[i];
        s.p1_valid = false;
        s.in_off = at; s.in_len = lens[i];
        at += lens[i];
        (void)hipSetDevice(s.device);
        s.rc = zdw_set_input_device(s.ctx, d_pieces[i], lens[i]);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    }
    m->image_len = at;
    return first_failure(m, "taking the input");
}

// One block of the input (ConvertToZDW.cpp:765-894 is the reference's loop over them).  The shards that still hold rows take part as
// requesters; every shard owns a key range of the block's dictionary.  Under a memory limit the dictionary exchange pauses in the
// middle: the owners say which occurrence of every string is the first node-wide, the shards walk the reference's arenas over
// their own first occurrences in row order, and the shard in which the limit trips cuts its part of the block back to that field
// (stringheap.cpp:74-86, memory.cpp:76-81, ConvertToZDW.cpp:338,404); the owners then drop what was first met beyond it.
int zdw_multi_encode(zdw_multi *m, zdw_multi_block *blk)
{
    if (!m || !blk) return ZDW_UNKNOWN_ERROR;
    memset(blk, 0, sizeof *blk);
    blk->longest_line = m->longest_line;
    if (m->finished) { blk->is_last_block = 1; return ZDW_OK; }
    if (m->encoded) return mfail(m, ZDW_UNKNOWN_ERROR, "zdw_multi_next has not been called for the block before");
    const size_t n = m->sh.size();
    const uint32_t nc = m->ncols;
    const bool limited = m->mem_limit_mb > 0.0;
    const size_t first = m->first;                          // the requesters: shards first .. last (those that still hold rows, as far as
    size_t last = n - 1;                                    // the block can reach)
    int rc = ZDW_OK;
    // Under a memory limit not every shard that still holds rows has to take part: the block ends where the dictionary has grown to the
    // limit, so the shards are taken in row order until their strings -- counted as if none were shared: an upper bound of what the
    // block's dictionary holds -- reach twice the limit.  If the limit then does not trip among them (they share more than
    // expected), twice as many take part in a second attempt; the shards' pass 1 is kept either way.
    size_t window = limited ? 1 : n;
    bool tripped = false;
    size_t end_shard = 0, bad_shard = n;
    uint64_t rows_before_bad = 0;
    uint32_t trip_row = 0, trip_fields = 0;
    uint64_t trip_row_start = 0, trip_gpos = ~0ull;
    std::vector<std::vector<uint64_t>> cnt, back;
    std::vector<uint64_t> sent;
    bool maybe = false;
    for (;; window *= 2) {
    last = std::min(n - 1, first + window - 1);

    // ---- pass 1 on every shard that still holds rows and has none yet for them: ALL of them at once, not only the window's (the
    // ---- shards beyond it keep theirs for the block they end up in; one round of parallel launches instead of one per doubling) ----
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        s.rc = 0;
        if (i < first || s.p1_valid) return;
        s.mc_valid = false;
        s.rc = zdw_block_pass1(s.ctx, nullptr, &s.p1);
        if (s.rc) s.err = zdw_last_error(s.ctx); else s.p1_valid = true;
    });
    rc = first_failure(m, "pass 1");
    if (rc) return rc;
    for (size_t i = first; i <= last; ++i)
        if (!m->sh[i].p1.is_last_block) return mfail(m, ZDW_OUT_OF_MEMORY, "shard %zu does not fit its device as one block: use more devices or the one-GPU path", i);
    // the first row with a wrong field count, if any (the shards do not fail on one: the block may end before it)
    bad_shard = n;
    rows_before_bad = 0;
    for (size_t i = first; i <= last; ++i) {
        if (m->sh[i].p1.bad_row) { bad_shard = i; break; }
        rows_before_bad += m->sh[i].p1.num_rows;
    }

    // ---- can the limit trip at all?  (all shards' strings as if none were shared: an upper bound) ----
    maybe = false;
    if (limited) {
        uint64_t entries = 0, bytes = 0;
        uint32_t max_len = 0;
        for (size_t i = first; i <= last; ++i) {
            Shard &s = m->sh[i];
            if (!s.mc_valid) {              // (kept while the shard's pass 1 is: every doubling of the window asks again)
                rc = zdw_shard_model_counts(s.ctx, &s.mc_entries, &s.mc_bytes, &s.mc_max_len);
                if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
                s.mc_valid = true;
            }
            entries += s.mc_entries; bytes += s.mc_bytes; max_len = std::max(max_len, s.mc_max_len);
        }
        const double ub = zdw_model_upper_bound_mb(entries, bytes, max_len);
        maybe = ub >= m->mem_limit_mb;
        if (last < n - 1 && ub < 2.0 * m->mem_limit_mb) continue;      // too few strings to be sure the block ends among these shards: more of them
    }

    // ---- dictionary, range-partitioned: the strings travel to the owners of their key ranges ----
    constexpr uint32_t SAMPLES = 256;
    std::vector<uint64_t> splitters;            // 2 words each, n - 1 of them
    {
        std::vector<std::pair<uint64_t, uint64_t>> all;
        for (size_t i = first; i <= last; ++i) {
            uint64_t cnt = 0;
            rc = zdw_shard_sort_uniques(m->sh[i].ctx, &cnt);
            if (rc) return mfail(m, rc, "%s", zdw_last_error(m->sh[i].ctx));
            std::vector<uint64_t> keys(2 * SAMPLES);
            rc = zdw_shard_sample_keys(m->sh[i].ctx, SAMPLES, keys.data());
            if (rc) return mfail(m, rc, "%s", zdw_last_error(m->sh[i].ctx));
            for (uint32_t k = 0; k < SAMPLES; ++k)
                if (!(keys[2 * k] == ~0ull && keys[2 * k + 1] == ~0ull)) all.emplace_back(keys[2 * k], keys[2 * k + 1]);      // "no string here" carries no information
        }
        if (all.empty()) all.emplace_back(~0ull, ~0ull);
        std::sort(all.begin(), all.end());
        for (size_t j = 1; j < n; ++j) {
            const auto &k = all[std::min(all.size() - 1, j * all.size() / n)];
            splitters.push_back(k.first); splitters.push_back(k.second);
        }
    }
    cnt.assign(n, std::vector<uint64_t>(n, 0));
    std::vector<std::vector<uint64_t>> nby(n, std::vector<uint64_t>(n, 0));
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        s.counts.assign(n, 0); s.nbytes.assign(n, 0);
        s.d_lens = s.d_bytes = s.d_pos = nullptr;
        s.rc = 0;
        if (i < first || i > last) return;      // no rows left here, or rows the block cannot reach: sends nothing
        s.rc = zdw_shard_partition(s.ctx, (uint32_t)n, splitters.data(), s.counts.data(), s.nbytes.data(), &s.d_lens, &s.d_bytes);
        if (!s.rc && maybe) s.rc = zdw_shard_positions(s.ctx, s.in_off, &s.d_pos);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    rc = first_failure(m, "partitioning the dictionary");
    if (rc) return rc;
    for (size_t q = 0; q < n; ++q) for (size_t r = 0; r < n; ++r) { cnt[q][r] = m->sh[q].counts[r]; nby[q][r] = m->sh[q].nbytes[r]; }
    std::vector<uint64_t> rc_cnt(n, 0), rc_by(n, 0);                  // what every owner receives, what every requester sent
    sent.assign(n, 0);
    for (size_t r = 0; r < n; ++r) for (size_t q = 0; q < n; ++q) { rc_cnt[r] += cnt[q][r]; rc_by[r] += nby[q][r]; sent[q] += cnt[q][r]; }
    {
        std::vector<const void *> src_l(n), src_b(n), src_p(n);
        std::vector<void *> dst_l(n), dst_b(n), dst_p(n);
        for (size_t i = 0; i < n; ++i) {
            Shard &s = m->sh[i];
            rc = grow(m, s.device, &s.d_recv_lens, &s.cap_recv_lens, rc_cnt[i] * 4 + 16); if (rc) return rc;
            rc = grow(m, s.device, &s.d_recv_bytes, &s.cap_recv_bytes, rc_by[i] + 32); if (rc) return rc;
            if (maybe) { rc = grow(m, s.device, &s.d_recv_pos, &s.cap_recv_pos, rc_cnt[i] * 8 + 16); if (rc) return rc; }
            src_l[i] = s.d_lens; src_b[i] = s.d_bytes; src_p[i] = s.d_pos; dst_l[i] = s.d_recv_lens; dst_b[i] = s.d_recv_bytes; dst_p[i] = s.d_recv_pos;
        }
        rc = all_to_all_v(m, src_l, dst_l, cnt, 4); if (rc) return rc;
        rc = all_to_all_v(m, src_b, dst_b, nby, 1); if (rc) return rc;
        if (maybe) { rc = all_to_all_v(m, src_p, dst_p, cnt, 8); if (rc) return rc; }
    }
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        s.rc = zdw_owner_import(s.ctx, s.d_recv_lens, s.d_recv_bytes, maybe ? s.d_recv_pos : nullptr, rc_cnt[i], rc_by[i]);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    rc = first_failure(m, "building a dictionary range");
    if (rc) return rc;
    // what travels back to the requesters, owner r -> requester q: cnt[q][r] words in the order q sent its strings (range by range)
    back.assign(n, std::vector<uint64_t>(n));
    for (size_t r = 0; r < n; ++r) for (size_t q = 0; q < n; ++q) back[r][q] = cnt[q][r];
    auto words_back = [&](const std::vector<const void *> &src) -> int {
        std::vector<void *> dst(n);
        for (size_t i = 0; i < n; ++i) {
            Shard &s = m->sh[i];
            const int r2 = grow(m, s.device, &s.d_back, &s.cap_back, sent[i] * 4 + 16);
            if (r2) return r2;
            dst[i] = s.d_back;
        }
        return all_to_all_v(m, src, dst, back, 4);
    };

    // ---- where does the block end? ----
    end_shard = last;                           // the shard that holds the block's last row
    tripped = false;
    if (maybe) {
        std::vector<const void *> src(n);
        for_shards(m, [&](size_t i) {
            Shard &s = m->sh[i];
            s.d_offsets = nullptr;
            s.rc = zdw_owner_first_flags(s.ctx, &s.d_offsets);
            if (s.rc) s.err = zdw_last_error(s.ctx);
        });
        rc = first_failure(m, "first occurrences");
        if (rc) return rc;
        for (size_t i = 0; i < n; ++i) src[i] = m->sh[i].d_offsets;
        rc = words_back(src);
        if (rc) return rc;
        zdw_model_state st = {0, 0, 0};
        for (size_t i = first; i <= last && !tripped; ++i) {            // in row order: the state goes from shard to shard
            Shard &s = m->sh[i];
            uint32_t found = 0;
            uint64_t pos = 0;
            (void)hipSetDevice(s.device);
            rc = zdw_shard_model_walk(s.ctx, s.d_back, sent[i], m->mem_limit_mb, &st, &found, &trip_row, &trip_fields, &trip_row_start, &pos);
            if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
            if (found) { tripped = true; end_shard = i; trip_gpos = s.in_off + pos; }
        }
    }
    if (tripped || last == n - 1) break;       // the block ends among these shards, or takes all the rows that are left
    }                                           // (else: the limit does not trip among them: twice as many shards)
    auto words_back = [&](const std::vector<const void *> &src) -> int {
        std::vector<void *> dst(n);
        for (size_t i = 0; i < n; ++i) {
            Shard &s = m->sh[i];
            const int r2 = grow(m, s.device, &s.d_back, &s.cap_back, sent[i] * 4 + 16);
            if (r2) return r2;
            dst[i] = s.d_back;
        }
        return all_to_all_v(m, src, dst, back, 4);
    };
    if (bad_shard < n) {
        // the reference meets the rows in order: it reports the bad row unless the block ends before it (ConvertToZDW.cpp:336-338)
        const uint32_t bad0 = m->sh[bad_shard].p1.bad_row - 1;
        const bool reached = !tripped || bad_shard < end_shard || (bad_shard == end_shard && bad0 <= trip_row);
        if (reached) {
            blk->bad_row = (uint32_t)(rows_before_bad + m->sh[bad_shard].p1.bad_row);
            for (size_t i = first; i <= last; ++i) m->sh[i].p1_valid = false;
            return mfail(m, ZDW_WRONG_NUM_OF_COLUMNS_ON_A_ROW, "Row %u had the problem", blk->bad_row);
        }
    }
    if (tripped) {
        Shard &s = m->sh[end_shard];
        (void)hipSetDevice(s.device);
        rc = zdw_shard_trim(s.ctx, trip_row, trip_fields, trip_row_start, &s.p1);
        if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
    }
    uint64_t rows = 0;
    for (size_t i = first; i <= end_shard; ++i) rows += m->sh[i].p1.num_rows;
    if (rows >= 0xFFFFFFFFull) return mfail(m, ZDW_UNKNOWN_ERROR, "more than 2^32-2 rows in one block");
    if (tripped && rows == 0) return mfail(m, ZDW_OUT_OF_MEMORY, "block limit leaves no room for a single row");      // ConvertToZDW.cpp:824-835

    // ---- 1. column ranges of the block's shards: all-reduce of [~min, max, seen] with max (~x reverses the unsigned order) ----
    {
        std::vector<std::vector<uint64_t>> host(n, std::vector<uint64_t>(3 * (size_t)nc + 1, 0));
        for (size_t i = first; i <= end_shard; ++i) {
            Shard &s = m->sh[i];
            std::vector<uint64_t> mn(nc + 1), mx(nc + 1);
            std::vector<uint8_t> seen(nc + 1);
            rc = zdw_shard_export_ranges(s.ctx, mn.data(), mx.data(), seen.data());
            if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
            for (uint32_t k = 0; k < nc; ++k) { host[i][k] = ~mn[k]; host[i][nc + k] = mx[k]; host[i][2 * (size_t)nc + k] = seen[k]; }
        }
        const size_t cntw = 3 * (size_t)nc;
        if (cntw && n > 1) {
            for (size_t i = 0; i < n; ++i) {
                Shard &s = m->sh[i];
                size_t cap = s.d_ranges ? cntw * 8 : 0;
                if (!s.d_ranges) { rc = grow(m, s.device, &s.d_ranges, &cap, cntw * 8); if (rc) return rc; }
                M_HIP(m, hipSetDevice(s.device));
                M_HIP(m, hipMemcpyAsync(s.d_ranges, host[i].data(), cntw * 8, hipMemcpyHostToDevice, s.stream));
            }
            if (m->use_rccl) {
                M_NCCL(m, rccl().GroupStart());
                for (size_t i = 0; i < n; ++i) M_NCCL(m, rccl().AllReduce(m->sh[i].d_ranges, m->sh[i].d_ranges, cntw, ncclUint64, ncclMax, m->sh[i].comm, m->sh[i].stream));
                M_NCCL(m, rccl().GroupEnd());
                for (size_t i = 0; i < n; ++i) {
                    M_HIP(m, hipSetDevice(m->sh[i].device));
                    M_HIP(m, hipMemcpyAsync(host[i].data(), m->sh[i].d_ranges, cntw * 8, hipMemcpyDeviceToHost, m->sh[i].stream));
                    M_HIP(m, hipStreamSynchronize(m->sh[i].stream));
                }
            } else {
                for (size_t i = 0; i < n; ++i) { M_HIP(m, hipSetDevice(m->sh[i].device)); M_HIP(m, hipStreamSynchronize(m->sh[i].stream)); }
                for (size_t k = 0; k < cntw; ++k) {
                    uint64_t v = 0;
                    for (size_t i = 0; i < n; ++i) v = std::max(v, host[i][k]);
                    for (size_t i = 0; i < n; ++i) host[i][k] = v;
                }
            }
        }
        for (size_t i = first; i <= end_shard; ++i) {
            std::vector<uint64_t> mn(nc + 1), mx(nc + 1);
            std::vector<uint8_t> seen(nc + 1);
            for (uint32_t k = 0; k < nc; ++k) { mn[k] = ~host[i][k]; mx[k] = host[i][nc + k]; seen[k] = host[i][2 * (size_t)nc + k] ? 1 : 0; }
            rc = zdw_shard_import_ranges(m->sh[i].ctx, mn.data(), mx.data(), seen.data());
            if (rc) return mfail(m, rc, "%s", zdw_last_error(m->sh[i].ctx));
        }
    }

    // ---- 2. the owners sort and lay out what belongs to the block; every string's offset travels back ----
    // (When no string is dropped -- the block takes everything its shards sent -- the dictionary's size is known before the owners
    // sort: the sum of what their deduplicated ranges take.  The requesters then start their row lengths beside that sort.)
    uint64_t early_size = 0;
    if (!tripped) {
        early_size = 1;
        for (size_t i = 0; i < n && early_size; ++i) {
            uint64_t b = 0;
            if (zdw_owner_size(m->sh[i].ctx, &b) != ZDW_OK) early_size = 0; else early_size += b;
        }
        if (early_size >= (1ull << 32)) early_size = 0;
    }
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        if (early_size && i >= first && i <= end_shard) (void)zdw_shard_rows_early(s.ctx, early_size);      // (optional: a failure only means the ordinary order)
        s.rc = zdw_owner_finish(s.ctx, tripped ? trip_gpos : ~0ull, &s.slice);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    rc = first_failure(m, "building a dictionary range");
    if (rc) return rc;
    uint64_t dict_size = 1;                     // offset 0 is the origin byte (dictionary.cpp:96)
    std::vector<uint64_t> base(n);
    for (size_t i = 0; i < n; ++i) { base[i] = dict_size; dict_size += m->sh[i].slice; }
    if (dict_size >= (1ull << 32)) return mfail(m, ZDW_OUT_OF_MEMORY, "the block's dictionary exceeds 2^32 bytes: it has to be split (a memory limit does that)");
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        s.rc = zdw_owner_offsets(s.ctx, base[i], &s.d_offsets);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    rc = first_failure(m, "dictionary offsets");
    if (rc) return rc;
    {
        std::vector<const void *> src(n);
        for (size_t i = 0; i < n; ++i) src[i] = m->sh[i].d_offsets;
        rc = words_back(src);
        if (rc) return rc;
    }
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        // (a shard without rows in this block still owns its key range: its piece of the dictionary section)
        if (i >= first && i <= end_shard) s.rc = zdw_shard_set_offsets(s.ctx, s.d_back, sent[i], dict_size, i == 0 ? 1 : 0);
        else s.rc = zdw_owner_section(s.ctx, dict_size, i == 0 ? 1 : 0);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    rc = first_failure(m, "dictionary offsets");
    if (rc) return rc;

    // ---- 3. the row halo and the line buffer, chained over the block's shards in row order ----
    {
        std::vector<uint64_t> cur(nc + 1, 0);
        bool have = false;
        uint32_t cap = m->longest_line;                     // (it never shrinks, across blocks either: getnextrow.cpp:57-65)
        for (size_t i = first; i <= end_shard; ++i) {
            Shard &s = m->sh[i];
            rc = zdw_shard_set_prev_row(s.ctx, have ? cur.data() : nullptr);
            if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
            if (s.p1.num_rows) {
                rc = zdw_shard_last_row(s.ctx, cur.data());
                if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
                have = true;
            }
            uint32_t caps[ZDW_LINE_CAPS];
            rc = zdw_shard_line_caps(s.ctx, caps);
            if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
            uint32_t k = 0;
            while ((16384u << k) < cap && k + 1 < ZDW_LINE_CAPS) ++k;
            cap = caps[k];
        }
        m->longest_line = cap;
    }

    // ---- the shards' rows ----
    for_shards(m, [&](size_t i) {
        Shard &s = m->sh[i];
        s.rc = 0;
        if (i < first || i > end_shard) return;
        s.rc = zdw_block_pass2_device(s.ctx, nullptr, nullptr);
        if (s.rc) s.err = zdw_last_error(s.ctx);
    });
    rc = first_failure(m, "pass 2");
    if (rc) return rc;
    blk->num_rows = (uint32_t)rows;
    blk->longest_line = m->longest_line;
    blk->is_last_block = tripped ? 0 : 1;
    blk->dict_bytes = dict_size;
    m->blk_first = first; m->blk_end = end_shard; m->blk_tripped = tripped;
    m->encoded = true;
    return ZDW_OK;
}

int zdw_multi_next(zdw_multi *m)
{
    if (!m) return ZDW_UNKNOWN_ERROR;
    if (!m->encoded) return ZDW_OK;
    m->encoded = false;
    if (!m->blk_tripped) { m->finished = true; return ZDW_OK; }
    // the shard that held the block's end goes on behind it; the shards before it are done, those behind it keep their pass 1
    Shard &s = m->sh[m->blk_end];
    (void)hipSetDevice(s.device);
    const int rc = zdw_block_next(s.ctx);
    if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
    s.p1_valid = false;
    m->first = m->blk_end;
    return ZDW_OK;
}

int zdw_multi_stats(zdw_multi *m, uint8_t *col_size, uint64_t *col_base, uint32_t *used)
{
    if (!m || !m->encoded) return m ? mfail(m, ZDW_UNKNOWN_ERROR, "zdw_multi_encode has not run") : ZDW_UNKNOWN_ERROR;
    Shard &s0 = m->sh[m->blk_first];                                             // the same on every shard of the block (merged ranges, one dictionary size)
    (void)hipSetDevice(s0.device);
    const int rc = zdw_block_stats(s0.ctx, col_size, col_base, used);
    if (rc) return mfail(m, rc, "%s", zdw_last_error(s0.ctx));
    return ZDW_OK;
}

int zdw_multi_dict_write(zdw_multi *m, zdw_write_fn write, void *user, uint64_t *section_len)
{
    if (!m || !m->encoded || !write) return m ? mfail(m, ZDW_UNKNOWN_ERROR, "zdw_multi_encode has not run") : ZDW_UNKNOWN_ERROR;
    uint64_t total = 0;
    for (Shard &s : m->sh) {            // the pieces of ranks 0 .. N-1 concatenate to the reference's section
        uint64_t len = 0;
        (void)hipSetDevice(s.device);
        const int rc = zdw_block_dict_write(s.ctx, write, user, &len, nullptr);
        if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
        total += len;
    }
    if (section_len) *section_len = total;
    return ZDW_OK;
}

int zdw_multi_rows_write(zdw_multi *m, zdw_write_fn write, void *user, uint64_t *rows_len)
{
    if (!m || !m->encoded || !write) return m ? mfail(m, ZDW_UNKNOWN_ERROR, "zdw_multi_encode has not run") : ZDW_UNKNOWN_ERROR;
    uint64_t total = 0;
    for (size_t i = m->blk_first; i <= m->blk_end; ++i) {
        Shard &s = m->sh[i];
        uint64_t len = 0;
        (void)hipSetDevice(s.device);
        const int rc = zdw_block_pass2_write(s.ctx, write, user, &len);
        if (rc) return mfail(m, rc, "%s", zdw_last_error(s.ctx));
        total += len;
    }
    if (rows_len) *rows_len = total;
    return ZDW_OK;
}

}  // extern "C"
