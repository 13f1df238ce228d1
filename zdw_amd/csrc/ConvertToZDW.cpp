// ConvertToZDW.cpp — host shim of the MI355X encoder: the reference's file-level behaviour
// (cplusplus/ConvertToZDW.cpp) around GPU block encoding through include/zdw_hip.h.
//
// What stays here, with its reference anchor:
//   schema parsing ............. ReadDescFile        ConvertToZDW.cpp:91-162
//   metadata sidecar / checks .. loadMetadataFile    :240-262, validateMetadata :226-237
//   file header ................ processFile         :673-737
//   block header ............... processFile         :839-842
//   compressor child, temp name, rename ............ :647-671, :926-944
//   validation shell-out ....... validate            :166-223
// What moved to the GPU: parseInput, Dictionary, writeLookupColumnStats' inputs, writeBlockRows.
#include "ConvertToZDW.h"

#include "zdw_hip.h"
#include "parallel_gzip.h"

#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>
#include <strings.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <fcntl.h>
#include <unistd.h>

namespace adobe {
namespace zdw {

void defaultStatusOutputCallback(const StatusOutputLevel level, const char *format, ...)
{
    FILE *to = level == ERROR ? stderr : stdout;
    va_list ap;
    va_start(ap, format);
    vfprintf(to, format, ap);
    va_end(ap);
    fflush(to);
}

void stdErrStatusOutputCallback(const StatusOutputLevel, const char *format, ...)
{
    va_list ap;
    va_start(ap, format);
    vfprintf(stderr, format, ap);
    va_end(ap);
    fflush(stderr);
}

const int ConvertToZDW::CONVERT_ZDW_CURRENT_VERSION = 11;
const char ConvertToZDW::CONVERT_ZDW_VERSION_TAIL[3] = "b";

const char ConvertToZDW::ERR_CODE_TEXTS[ERR_CODE_COUNT][30] = {
    "OK", "NO_ARGS", "CONVERSION_FAILED", "UNTAR_FAILED", "MISSING_DESC_FILE", "MISSING_SQL_FILE",
    "FILE_CREATION_ERR", "OUT_OF_MEMORY", "UNCONVERT_FAILED", "FILE_SIZES_DIFFER",
    "FILES_DIFFER", "MISSING_ARGUMENT", "GZIP_FAILED", "BZIP2_FAILED",
    "DESC_FILE_MISSING_TYPE_INFO", "WRONG_NUM_OF_COLUMNS_ON_A_ROW",
    "BAD_PARAMETER", "TOO_MANY_INPUT_FILES", "NO_INPUT_FILES", "CANT_OPEN_TEMP_FILE",
    "Unknown error",
    "BAD_METADATA_PARAMETER", "BAD_METADATA_FILE"};

ConvertToZDW::ConvertToZDW(const bool quiet, const bool streaming)
    : compressor(GZIP), statusOutput(defaultStatusOutputCallback), bQuiet(quiet), bTrimTrailingSpaces(false),
      bStreamingInput(streaming), device(-1), formatVersion(CONVERT_ZDW_CURRENT_VERSION), memLimitMB(3072.0)
{
}

ConvertToZDW::~ConvertToZDW() {}

bool ConvertToZDW::setFormatVersion(int version)
{
    if (version < 9 || version > CONVERT_ZDW_CURRENT_VERSION) return false;
    formatVersion = version;
    return true;
}

void ConvertToZDW::setBlockSplits(const std::vector<uint32_t> &rows, const std::vector<uint32_t> &fields)
{
    splitRows = rows;
    splitFields = fields;
    splitFields.resize(splitRows.size(), 0);
}

const char *ConvertToZDW::compressorExtension() const
{
    switch (compressor) {
    case GZIP: return ".gz";
    case BZIP2: return ".bz2";
    case XZ: case FXZ: return ".xz";
    case ZSTD: return ".zst";
    }
    return "";
}

const char *ConvertToZDW::compressorCommand() const
{
    switch (compressor) {
    case GZIP: return "gzip";
    case BZIP2: return "bzip2";
    case XZ: return "xz";
    case FXZ: return "fxz";
    case ZSTD: return "zstd";
    }
    return "";
}

static bool startsWith(const char *s, const char *prefix) { return strncmp(s, prefix, strlen(prefix)) == 0; }

// One schema line is "<name>\t<sql type ...>"; lines are consumed in 1023-byte pieces like the
// reference's fgets buffer, so over-long lines split the same way.
unsigned ConvertToZDW::readDescFile(FILE *f, Schema &s) const
{
    enum : unsigned char { VARCHAR = 0, TEXT = 1, DATETIME = 2, CHAR_2 = 3, CHAR = 6, TINY = 7, SHORT = 8, LONG = 9, LONGLONG = 10,
                           DECIMAL = 11, TINY_SIGNED = 12, SHORT_SIGNED = 13, LONG_SIGNED = 14, LONGLONG_SIGNED = 15,
                           TINYTEXT = 16, MEDIUMTEXT = 17, LONGTEXT = 18 };
    s.names.clear(); s.types.clear(); s.charSizes.clear();
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        if (strncasecmp(line, "Field", 5) == 0) continue;
        char *type = strchr(line, '\t');
        if (!type) return BAD_FIELD;
        *type++ = '\0';
        s.names.push_back(line);
        int charSize = 0;
        unsigned char id;
        if (startsWith(type, "varchar")) { id = VARCHAR; charSize = atoi(type + 8); }
        else if (startsWith(type, "char")) {
            charSize = atoi(type + 5);
            id = charSize == 1 ? CHAR : charSize == 2 ? CHAR_2 : VARCHAR;
        }
        else if (startsWith(type, "text")) id = TEXT;
        else if (startsWith(type, "tinytext")) id = TINYTEXT;
        else if (startsWith(type, "mediumtext")) id = MEDIUMTEXT;
        else if (startsWith(type, "longtext")) id = LONGTEXT;
        else if (startsWith(type, "datetime")) id = DATETIME;
        else if (startsWith(type, "decimal") || (type[0] && startsWith(type + 1, "decimal"))) id = DECIMAL;
        else {
            const bool isSigned = strstr(type, "unsigned") == NULL;
            if (startsWith(type, "tinyint")) id = isSigned ? TINY_SIGNED : TINY;
            else if (startsWith(type, "smallint")) id = isSigned ? SHORT_SIGNED : SHORT;
            else if (startsWith(type, "bigint")) id = isSigned ? LONGLONG_SIGNED : LONGLONG;
            else id = isSigned ? LONG_SIGNED : LONG;
        }
        s.types.push_back(id);
        s.charSizes.push_back(charSize);
    }
    return static_cast<unsigned>(s.names.size());
}

bool ConvertToZDW::validateMetadata(const std::map<std::string, std::string> &metadata) const
{
    for (const auto &kv : metadata) {
        if (kv.first.find_first_of("=\n") != std::string::npos) return false;
        if (kv.second.find('\n') != std::string::npos) return false;
    }
    return true;
}

int ConvertToZDW::loadMetadataFile(const char *filepath, std::map<std::string, std::string> &metadata)
{
    std::ifstream in(filepath);
    if (!in) return -1;
    std::string line;
    int lineNo = 0;
    while (std::getline(in, line)) {
        ++lineNo;
        if (line.empty()) continue;
        const size_t eq = line.find('=');
        if (eq == std::string::npos) return lineNo;
        metadata[line.substr(0, eq)] = line.substr(eq + 1);
    }
    return 0;
}

ConvertToZDW::ERR_CODE ConvertToZDW::validate(const char *zdwFile, const std::vector<std::string> &srcFiles, const char *exeName,
                                              const char *outputDir)
{
    if (!bQuiet) statusOutput(INFO, "Unconverting %s back for validation...\n", zdwFile);
    // the decoder is expected next to this executable, as in the reference
    std::string dir(exeName);
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
    // (ZDW_VALIDATE_GPU=1: the sibling decodes on the device -- unconvertDWfile --gpu -- instead of on one host core)
    std::string unconvert = dir + (getenv("ZDW_VALIDATE_GPU") ? "unconvertDWfile --gpu -q - " : "unconvertDWfile -q - ");
    if (outputDir) { unconvert += "-d "; unconvert += outputDir; unconvert += " "; }
    unconvert += zdwFile;
    std::string cmd;
    if (bStreamingInput) {
        std::string zcat = "zcat ";
        for (const std::string &f : srcFiles) { zcat += f; zcat += " "; }
        cmd = "/bin/bash -c \"cmp <(" + unconvert + ") <(" + zcat + ")\"";
    } else if (bTrimTrailingSpaces) {
        cmd = "/bin/bash -c \"cmp <(" + unconvert + ") <(" + dir + "trim_spaces " + srcFiles[0] + ")\"";
    } else {
        cmd = unconvert + " | cmp " + srcFiles[0];
    }
    if (!bQuiet) statusOutput(INFO, "VALIDATION COMMAND: %s\n", cmd.c_str());
    return system(cmd.c_str()) == 0 ? OK : FILES_DIFFER;
}

namespace {

// The compressor's pipe behind a writer thread: put() copies into a queue of at most QUEUE_LIMIT bytes and returns, the thread
// writes.  So the next piece of a block leaves the device while the one before goes down the pipe, and the next block's pass 1
// runs while the last pieces of this one drain (the reference writes and parses in turn).
struct OutPipe {
    FILE *f = nullptr;                                       // the compressor child's stdin, or (gz) the output file itself
    std::unique_ptr<zdwhost::ParallelGzip> gz;               // gzip in this process, on several threads (parallel_gzip.h)
    static constexpr size_t QUEUE_LIMIT = 512u << 20;
    std::thread th;
    std::mutex m;
    std::condition_variable can_put, can_get;
    std::deque<std::vector<unsigned char> > q;
    std::vector<std::vector<unsigned char> > spare;          // emptied pieces, their memory kept
    size_t queued = 0;
    bool closing = false, failed = false, started = false;

    void run()
    {
        for (;;) {
            std::vector<unsigned char> piece;
            {
                std::unique_lock<std::mutex> lk(m);
                can_get.wait(lk, [&] { return closing || !q.empty(); });
                if (q.empty()) return;
                piece.swap(q.front());
                q.pop_front();
            }
            const bool ok = failed || (gz ? gz->write(piece.data(), piece.size())                   // (after a failure: drain without writing)
                                          : fwrite(piece.data(), 1, piece.size(), f) == piece.size());
            {
                std::lock_guard<std::mutex> g(m);
                queued -= piece.size();
                if (!ok) failed = true;
                if (spare.size() < 4) { piece.clear(); spare.push_back(std::move(piece)); }
            }
            can_put.notify_all();
        }
    }
    bool put(const void *p, size_t n)
    {
        if (n == 0) return true;
        if (!started) { started = true; th = std::thread([this] { run(); }); }
        std::vector<unsigned char> piece;
        {
            std::unique_lock<std::mutex> lk(m);
            if (failed) return false;
            can_put.wait(lk, [&] { return queued == 0 || queued + n <= QUEUE_LIMIT; });
            if (!spare.empty()) { piece.swap(spare.back()); spare.pop_back(); }
            queued += n;
        }
        // small writes (headers, column statistics) ride on the piece before them when it has not left yet
        piece.assign(static_cast<const unsigned char *>(p), static_cast<const unsigned char *>(p) + n);
        {
            std::lock_guard<std::mutex> g(m);
            if (n < 4096 && !q.empty() && q.back().size() < (1u << 20)) q.back().insert(q.back().end(), piece.begin(), piece.end());
            else q.push_back(std::move(piece));
        }
        can_get.notify_one();
        return true;
    }
    bool le(uint64_t v, unsigned bytes)
    {
        unsigned char b[8];
        for (unsigned i = 0; i < 8; ++i) b[i] = static_cast<unsigned char>(v >> (8 * i));
        return put(b, bytes);
    }
    // everything queued is written (or dropped after a failure); returns false when a write failed
    bool finish()
    {
        if (started) {
            { std::lock_guard<std::mutex> g(m); closing = true; }
            can_get.notify_all();
            th.join();
            started = false;
        }
        return !failed;
    }
    // Through `cmd` (a shell command that reads the stream on its stdin and writes the file), or with gzThreads > 0 deflated here into `path`
    bool open(const std::string &cmd, const std::string &path, unsigned gzThreads, int gzLevel)
    {
        if (gzThreads) {
            f = fopen(path.c_str(), "wb");
            if (f) gz.reset(new zdwhost::ParallelGzip(f, gzThreads, gzLevel));
        } else {
            f = popen(cmd.c_str(), "w");
        }
        return f != nullptr;
    }
    // after finish(): the child's exit status (0: fine), or that of the writer in this process
    int close()
    {
        if (!f) return 0;
        int st;
        if (gz) {
            st = gz->close() ? 0 : 1;
            gz.reset();
            if (fclose(f) != 0) st = 1;
        } else {
            st = pclose(f);
        }
        f = nullptr;
        return st;
    }
    ~OutPipe() { finish(); if (f) (void)close(); }
};

// gzip without arguments the child would need to see (or with a plain level): deflated in this process on several threads, unless
// ZDW_GZIP_THREADS=0 asks for the reference's child process.  Returns the number of threads (0: the child), *level for them.
static unsigned inProcessGzip(bool isGzip, const char *zArgs, int *level)
{
    *level = 6;
    if (!isGzip) return 0;
    if (zArgs && *zArgs) {
        const std::string a = zArgs;
        if (a.size() == 2 && a[0] == '-' && a[1] >= '1' && a[1] <= '9') *level = a[1] - '0';
        else if (a == "--fast") *level = 1;
        else if (a == "--best") *level = 9;
        else return 0;
    }
    unsigned n = std::thread::hardware_concurrency();
    if (n > 16) n = 16;                                      // (a GPU's share of the host's cores)
    if (n < 1) n = 1;
    if (const char *e = getenv("ZDW_GZIP_THREADS")) n = (unsigned)atoi(e);
    return n > 64 ? 64 : n;
}

// The input read from a descriptor by the library's feeder thread (zdw_set_input_stream); with -i -v the bytes are also teed into
// the temporary file the validation compares against (ConvertToZDW.cpp:786-803: the reference tees stdin the same way).
struct InSource {
    int fd = -1;
    FILE *tee = nullptr;
    bool teeFailed = false;
};
int64_t readSource(void *user, uint8_t *buf, uint64_t cap)
{
    InSource *src = static_cast<InSource *>(user);
    for (;;) {
        const ssize_t r = read(src->fd, buf, cap > (1u << 30) ? (1u << 30) : static_cast<size_t>(cap));
        if (r < 0 && errno == EINTR) continue;
        if (r > 0 && src->tee && fwrite(buf, 1, static_cast<size_t>(r), src->tee) != static_cast<size_t>(r)) src->teeFailed = true;
        return static_cast<int64_t>(r);
    }
}

// ZDW_CLI_TIMING=1: wall time of the host-side phases of one file, on stderr
struct PhaseClock {
    const bool on = getenv("ZDW_CLI_TIMING") != nullptr;
    std::chrono::steady_clock::time_point last = std::chrono::steady_clock::now();
    std::vector<std::pair<const char *, double> > acc;
    void mark(const char *what)
    {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        const double ms = std::chrono::duration<double, std::milli>(now - last).count();
        last = now;
        for (auto &e : acc) if (!strcmp(e.first, what)) { e.second += ms; return; }
        acc.push_back(std::make_pair(what, ms));
    }
    ~PhaseClock()
    {
        if (!on) return;
        double tot = 0;
        for (auto &e : acc) tot += e.second;
        fprintf(stderr, "[zdw timing]");
        for (auto &e : acc) fprintf(stderr, " %s=%.1fms", e.first, e.second);
        fprintf(stderr, " total=%.1fms\n", tot);
    }
};

// zdw_write_fn for the streamed outputs: straight into the compressor's pipe
int writeToPipe(void *user, const uint8_t *bytes, uint64_t len)
{
    OutPipe *out = static_cast<OutPipe *>(user);
    return out->put(bytes, static_cast<size_t>(len)) ? 0 : 1;
}

struct CtxGuard {
    zdw_ctx *ctx = nullptr;
    ~CtxGuard() { if (ctx) zdw_ctx_destroy(ctx); }
};

}  // namespace

ConvertToZDW::ERR_CODE ConvertToZDW::processFile(const std::vector<unsigned char> *hostImage, const char *mappedImage, uint64_t imageLen,
                                                 int streamFd, const Schema &schema, const char *filestub, bool bValidate, const char *exeName,
                                                 const char *outputDir, const char *zArgs,
                                                 const std::map<std::string, std::string> &metadata)
{
    if (!validateMetadata(metadata)) {
        statusOutput(ERROR, "Invalid metadata parameter\n");
        return BAD_METADATA_PARAM;
    }
    const size_t numColumns = schema.names.size();
    // what an older format version cannot hold
    if (formatVersion < 11 && !metadata.empty()) {
        statusOutput(ERROR, "File metadata needs format version 11 (writing version %d)\n", formatVersion);
        return BAD_METADATA_PARAM;
    }
    if (formatVersion < 10) {
        for (size_t c = 0; c < numColumns; ++c)
            if (schema.types[c] == 17 || schema.types[c] == 18) {      // mediumtext, longtext
                statusOutput(ERROR, "Column %s needs format version 10 (writing version %d)\n", schema.names[c].c_str(), formatVersion);
                return BAD_PARAMETER;
            }
    }

    // ---- output names: <dir>/<base>.zdw<ext>, written as <base>.creating.zdw<ext> first ----
    std::string base;
    if (!outputDir) base = filestub;
    else {
        const char *leaf = strrchr(filestub, '/');
        base = std::string(outputDir) + "/" + (leaf ? leaf + 1 : filestub);
    }
    const std::string finalName = base + ".zdw" + compressorExtension();
    const std::string tempName = base + ".creating.zdw" + compressorExtension();

    PhaseClock clock;
    const bool several = devices.size() > 1 || (devices.size() == 1 && getenv("ZDW_MULTI_ALWAYS"));
    // (declared before the context: its feeder thread reads through `source` until the context is destroyed)
    InSource source;
    std::string teeName;
    struct TeeGuard { InSource &s; ~TeeGuard() { if (s.tee) pclose(s.tee); } } teeGuard{source};
    // ---- GPU context first: nothing is created on disk when no device is usable ----
    CtxGuard guard;
    struct MultiGuard { zdw_multi *m = NULL; ~MultiGuard() { if (m) zdw_multi_destroy(m); } } mguard;
    const unsigned char *image = hostImage ? hostImage->data() : reinterpret_cast<const unsigned char *>(mappedImage);
    if (several) {
        // one block after the other over several devices (zdw_multi_*): every block row-sharded, --mem-limit ends them where the
        // reference's end (the decision is taken on the node-wide dictionary)
        zdw_opts opts;
        memset(&opts, 0, sizeof opts);
        opts.struct_size = sizeof opts;
        opts.trim_trailing_spaces = bTrimTrailingSpaces ? 1u : 0u;
        opts.mem_limit_mb = memLimitMB > 0.0 ? static_cast<uint64_t>(memLimitMB) : 0;
        std::vector<int32_t> devs(devices.begin(), devices.end());
        int rc = zdw_multi_create(schema.types.data(), static_cast<uint32_t>(numColumns), &opts, devs.data(), static_cast<uint32_t>(devs.size()), &mguard.m);
        if (rc != ZDW_OK) {
            libError = zdw_multi_last_error(NULL);
            statusOutput(ERROR, "GPU encoder unavailable: %s\n", libError.c_str());
            return static_cast<ERR_CODE>(rc);
        }
        if (!splitRows.empty()) {
            statusOutput(ERROR, "ZDW_SPLIT_ROWS is not available with --gpus / --devices\n");
            return BAD_PARAMETER;
        }
        rc = zdw_multi_set_input_host(mguard.m, image, imageLen);
        if (rc != ZDW_OK) {
            libError = zdw_multi_last_error(mguard.m);
            statusOutput(ERROR, "%s\n", libError.c_str());
            return static_cast<ERR_CODE>(rc);
        }
    } else {
        zdw_opts opts;
        memset(&opts, 0, sizeof opts);
        opts.struct_size = sizeof opts;
        opts.device = device;
        opts.trim_trailing_spaces = bTrimTrailingSpaces ? 1u : 0u;
        if (const char *e = getenv("ZDW_DICT_SLOTS")) opts.dict_slots = strtoull(e, NULL, 10);
        if (const char *e = getenv("ZDW_MAX_DICT_BYTES")) opts.max_dict_bytes = strtoull(e, NULL, 10);
        if (const char *e = getenv("ZDW_MAX_DEVICE_BYTES")) opts.max_device_bytes = strtoull(e, NULL, 10);
        opts.mem_limit_mb = memLimitMB > 0.0 ? static_cast<uint64_t>(memLimitMB) : 0;
        const int rc = zdw_ctx_create(schema.types.data(), static_cast<uint32_t>(numColumns), &opts, &guard.ctx);
        if (rc != ZDW_OK) {
            libError = zdw_last_error(NULL);
            statusOutput(ERROR, "GPU encoder unavailable: %s\n", libError.c_str());
            return static_cast<ERR_CODE>(rc);
        }
    }
    zdw_ctx *ctx = guard.ctx;
    zdw_multi *mm = mguard.m;
    clock.mark("ctx_create");
    if (!several) {
        int rc;
        if (streamFd >= 0) {
            // the input is read while its blocks are encoded; with -i -v it is teed into <base>.tmp.1.gz on the way
            source.fd = streamFd;
            if (bValidate && bStreamingInput) {
                teeName = base + ".tmp.1.gz";
                source.tee = popen(("gzip > " + teeName).c_str(), "w");
                if (!source.tee) return CANT_OPEN_TEMP_FILE;
            }
            rc = zdw_set_input_stream(ctx, readSource, &source, imageLen);
        } else if (!hostImage && getenv("ZDW_CLI_STREAM_FILE")) {
            // a mapped file copied from the page cache piece by piece while its blocks are encoded.  Not the default: measured on a
            // 16 GB file of five blocks, one copy of everything (50 GB/s) followed by the encode (0.13 s) was no slower than the
            // pieces (33 GB/s, the blocks in between) -- the pipe to the compressor takes longer than both.
            rc = zdw_set_input_host_async(ctx, image, imageLen);
        } else {
            rc = zdw_set_input_host(ctx, image, imageLen);
        }
        if (rc != ZDW_OK) {
            libError = zdw_last_error(ctx);
            statusOutput(ERROR, "%s\n", libError.c_str());
            return static_cast<ERR_CODE>(rc);
        }
    }

    clock.mark("input_to_gpu");
    std::string cmd = compressorCommand();
    if (zArgs) { cmd += " "; cmd += zArgs; }
    cmd += " > ";
    cmd += tempName;
    OutPipe out;
    int gzLevel = 6;
    const unsigned gzThreads = inProcessGzip(compressor == GZIP, zArgs, &gzLevel);
    if (!out.open(cmd, tempName, gzThreads, gzLevel)) {
        statusOutput(ERROR, "Could not open the process '%s' for writing!\n", cmd.c_str());
        return FILE_CREATION_ERR;
    }

    // ---- file header ----
    bool ok = out.le(formatVersion, 2);
    if (formatVersion >= 11) {
        uint64_t metaLen = 0;
        for (const auto &kv : metadata) metaLen += kv.first.size() + kv.second.size() + 2;
        ok = ok && out.le(metaLen, 4);
        for (const auto &kv : metadata)
            ok = ok && out.put(kv.first.c_str(), kv.first.size() + 1) && out.put(kv.second.c_str(), kv.second.size() + 1);
    }
    for (size_t c = 0; c < numColumns; ++c) ok = ok && out.put(schema.names[c].c_str(), schema.names[c].size() + 1);
    ok = ok && out.le(0, 1);
    ok = ok && out.put(schema.types.data(), numColumns);
    for (size_t c = 0; c < numColumns; ++c) ok = ok && out.le(static_cast<uint16_t>(schema.charSizes[c]), 2);

    // ---- blocks ----
    ERR_CODE res = OK;
    uint32_t totalRows = 0;
    int blocks = 0;
    bool more = true;
    std::vector<unsigned char> colSize(numColumns ? numColumns : 1);
    std::vector<uint64_t> colBase(numColumns ? numColumns : 1);
    while (several && more && res == OK) {
        ++blocks;
        if (!bQuiet) {
            if (blocks == 1) statusOutput(INFO, "\nProcessing %s\n", filestub);
            else statusOutput(INFO, "\nProcessing block %d of %s (%llu rows so far)\n", blocks, filestub, (unsigned long long)totalRows);
        }
        zdw_multi_block blk;
        memset(&blk, 0, sizeof blk);
        int rc = zdw_multi_encode(mm, &blk);
        clock.mark("pass1+dict+pass2");
        if (rc == ZDW_WRONG_NUM_OF_COLUMNS_ON_A_ROW) {
            statusOutput(ERROR, "\nRow %u had the problem\n", blk.bad_row);
            res = WRONG_NUM_OF_COLUMNS_ON_A_ROW;
            break;
        }
        if (rc != ZDW_OK) {
            libError = zdw_multi_last_error(mm);
            if (rc == ZDW_OUT_OF_MEMORY && blk.num_rows == 0) statusOutput(ERROR, "Not enough memory to run %s\n", exeName);
            statusOutput(ERROR, "%s\n", libError.c_str());
            res = static_cast<ERR_CODE>(rc);
            break;
        }
        if (blk.num_rows == 0) {
            if (blocks == 1) statusOutput(ERROR, "Empty data file -- nothing to process\n");
            break;
        }
        if (!bQuiet) statusOutput(INFO, "\r%u rows\n", blk.num_rows);
        ok = ok && out.le(blk.num_rows, 4) && out.le(blk.longest_line, 4) && out.le(blk.is_last_block ? 1 : 0, 1);
        uint32_t used = 0;
        rc = zdw_multi_dict_write(mm, writeToPipe, &out, NULL);
        clock.mark("write_dict");
        if (rc == ZDW_OK) rc = zdw_multi_stats(mm, colSize.data(), colBase.data(), &used);
        if (rc == ZDW_OK) {
            ok = ok && out.put(colSize.data(), numColumns);
            for (uint32_t u = 0; u < used; ++u) ok = ok && out.le(colBase[u], 8);
            rc = zdw_multi_rows_write(mm, writeToPipe, &out, NULL);
            clock.mark("write_rows");
        }
        if (rc == ZDW_OK) rc = zdw_multi_next(mm);
        if (rc != ZDW_OK) {
            libError = zdw_multi_last_error(mm);
            statusOutput(ERROR, "%s\n", libError.c_str());
            res = static_cast<ERR_CODE>(rc);
            break;
        }
        totalRows += blk.num_rows;
        if (!bQuiet) statusOutput(INFO, "\r%u\nDone with block %d -- cleaning up...\n", blk.num_rows, blocks);
        more = !blk.is_last_block;
    }
    while (!several && more && res == OK) {
        ++blocks;
        if (!bQuiet) {
            if (blocks == 1) statusOutput(INFO, "\nProcessing %s\n", filestub);
            else statusOutput(INFO, "\nProcessing block %d of %s (%llu rows so far)\n", blocks, filestub, (unsigned long long)totalRows);
            statusOutput(INFO, "Compiling unique values\n");
        }
        zdw_split_hint hint = {0xFFFFFFFFu, 0};
        bool forced = static_cast<size_t>(blocks - 1) < splitRows.size();
        if (forced) { hint.max_rows = splitRows[blocks - 1]; hint.partial_fields = splitFields[blocks - 1]; }
        zdw_pass1_result p1;
        uint64_t sectionLen = 0; uint8_t indexSize = 0;
        int rc;
        // A block whose dictionary outgrows its limit, or whose outputs do not fit the device, is redone with half the
        // rows (the reference ends a block for the same reason -- memory -- at a point that depends on the machine).
        // Everything of the block is built on the device before a byte of it is written, so a retry costs no output.
        uint32_t used = 0;
        for (;;) {
            rc = zdw_block_pass1(ctx, forced ? &hint : NULL, &p1);
            if (rc != ZDW_OK || p1.num_rows == 0) break;
            rc = zdw_block_dict_device(ctx, NULL, &sectionLen, &indexSize);
            if (rc == ZDW_OK) rc = zdw_block_stats(ctx, colSize.data(), colBase.data(), &used);
            if (rc == ZDW_OK) rc = zdw_block_pass2_device(ctx, NULL, NULL);
            if (rc != ZDW_OUT_OF_MEMORY || p1.num_rows < 2) break;
            forced = true;
            hint.max_rows = p1.num_rows / 2;
            hint.partial_fields = 0;
            if (!bQuiet) statusOutput(INFO, "\nBlock too large for the device or its dictionary limit: retrying with %u rows\n", hint.max_rows);
        }
        clock.mark("pass1+dict+pass2");
        if (rc == ZDW_WRONG_NUM_OF_COLUMNS_ON_A_ROW) {
            statusOutput(ERROR, "\nRow %u had the problem\n", p1.bad_row);
            res = WRONG_NUM_OF_COLUMNS_ON_A_ROW;
            break;
        }
        if (rc != ZDW_OK) {
            libError = zdw_last_error(ctx);
            if (rc == ZDW_OUT_OF_MEMORY && p1.num_rows == 0) statusOutput(ERROR, "Not enough memory to run %s\n", exeName);
            else statusOutput(ERROR, "%s\n", libError.c_str());
            res = static_cast<ERR_CODE>(rc);
            break;
        }
        if (!bQuiet) statusOutput(INFO, "\r%u rows\n", p1.num_rows);
        if (p1.num_rows == 0) {
            // nothing (more) to encode; for a first block this is the empty-file case: header only
            if (blocks == 1) statusOutput(ERROR, "Empty data file -- nothing to process\n");
            break;
        }
        ok = ok && out.le(p1.num_rows, 4) && out.le(p1.longest_line, 4) && out.le(p1.is_last_block ? 1 : 0, 1);

        if (rc == ZDW_OK) {
            if (!bQuiet) {
                const uint64_t dictSize = indexSize ? sectionLen - 1 - indexSize : 1;
                statusOutput(INFO, "\nWriting dictionary:\n%u bytes being stored for %u unique entries.  Generating %d-byte offsets...\n",
                             static_cast<unsigned>(dictSize), static_cast<unsigned>(p1.dict_entries), indexSize ? indexSize : 1);
            }
            rc = zdw_block_dict_write(ctx, writeToPipe, &out, NULL, NULL);
            if (rc != ZDW_OK) ok = false;
            clock.mark("write_dict");
            if (rc == ZDW_OK) {
                ok = ok && out.put(colSize.data(), numColumns);
                for (uint32_t u = 0; u < used; ++u) ok = ok && out.le(colBase[u], 8);
                if (!bQuiet) statusOutput(INFO, "\nWriting rows\n");
                if (zdw_block_pass2_write(ctx, writeToPipe, &out, NULL) != ZDW_OK) ok = false;
                clock.mark("write_rows");
            }
        }
        if (rc != ZDW_OK) {
            libError = zdw_last_error(ctx);
            statusOutput(ERROR, "%s\n", libError.c_str());
            res = static_cast<ERR_CODE>(rc);
            break;
        }
        totalRows += p1.num_rows;
        if (!bQuiet) statusOutput(INFO, "\r%u\nDone with block %d -- cleaning up...\n", p1.num_rows, blocks);
        more = !p1.is_last_block;
        zdw_block_next(ctx);
    }

    ok = out.finish() && ok;
    const int childStatus = out.close();
    clock.mark("compressor_drain");
    if (res == OK && (!ok || childStatus != 0)) {
        statusOutput(ERROR, "Writing through '%s' failed\n", cmd.c_str());
        res = FILE_CREATION_ERR;
    }

    // ---- optional validation against the source text (needs the reference's unconvertDWfile beside us) ----
    if (res == OK && bValidate) {
        std::vector<std::string> src;
        std::string tmpGz;
        if (bStreamingInput && source.tee) {
            tmpGz = teeName;                                   // the copy of stdin made while it was read
            const int st = pclose(source.tee);
            source.tee = NULL;
            if (st != 0 || source.teeFailed) res = CANT_OPEN_TEMP_FILE;
            else src.push_back(tmpGz);
        } else if (bStreamingInput) {
            tmpGz = base + ".tmp.1.gz";
            FILE *gz = popen(("gzip > " + tmpGz).c_str(), "w");
            if (!gz) res = CANT_OPEN_TEMP_FILE;
            else {
                if (imageLen && fwrite(image, 1, imageLen, gz) != imageLen) res = CANT_OPEN_TEMP_FILE;
                pclose(gz);
                src.push_back(tmpGz);
            }
        } else {
            src.push_back(std::string(filestub) + "." + getInputFileExtension());
        }
        if (res == OK) {
            const ERR_CODE v = validate(tempName.c_str(), src, exeName, outputDir);
            if (v == OK) { if (!bQuiet) statusOutput(INFO, "%s GOOD\n", finalName.c_str()); }
            else { statusOutput(INFO, "%s BAD\n", finalName.c_str()); res = v; }
        }
        if (!tmpGz.empty()) unlink(tmpGz.c_str());
    } else if (!teeName.empty()) {
        // (the library's feeder thread may still be inside readSource(), writing to the tee -- a failed conversion does not
        // read its input to the end: the context goes first, which joins that thread, then the tee is closed)
        if (guard.ctx) { zdw_ctx_destroy(guard.ctx); guard.ctx = NULL; }
        if (source.tee) { pclose(source.tee); source.tee = NULL; }
        unlink(teeName.c_str());
    }

    if (res == OK) {
        if (!bQuiet) statusOutput(INFO, "Rows=%u\n", totalRows);    // parsed by callers of the reference CLI
        if (rename(tempName.c_str(), finalName.c_str()) == 0) {
            if (!bQuiet) statusOutput(INFO, "Done\n");
        } else {
            res = FILE_CREATION_ERR;
            statusOutput(INFO, "Final create file failed -- you can use %s instead.\n", tempName.c_str());
        }
    } else {
        unlink(tempName.c_str());
    }
    return res;
}

ConvertToZDW::ERR_CODE ConvertToZDW::convertFile(const char *infile, const char *exeName, const bool bValidate, char *filestub,
                                                 const char *outputDir, const char *zArgs,
                                                 const std::map<std::string, std::string> &metadata)
{
    // the stub is the input path cut at the first ".sql" 
    strcpy(filestub, infile);
    char *ext = strstr(filestub, ".sql");
    if (!ext) return MISSING_SQL_FILE;
    *ext = '\0';

    Schema schema;
    {
        const std::string descName = std::string(filestub) + ".desc." + getInputFileExtension();
        FILE *desc = fopen(descName.c_str(), "r");
        if (!desc) return MISSING_DESC_FILE;
        const unsigned n = readDescFile(desc, schema);
        fclose(desc);
        if (n == BAD_FIELD) return DESC_FILE_MISSING_TYPE_INFO;
    }

    std::map<std::string, std::string> meta = metadata;
    if (meta.empty()) {
        const std::string sidecar = std::string(filestub) + ".metadata";
        if (loadMetadataFile(sidecar.c_str(), meta) > 0) return BAD_METADATA_FILE;   // a missing sidecar is fine
    }

    if (const char *e = getenv("ZDW_SPLIT_ROWS")) {   // parity runs: force the block boundaries a reference run took
        std::vector<uint32_t> rows, fields;
        std::stringstream rs(e);
        std::string tok;
        while (std::getline(rs, tok, ',')) if (!tok.empty()) rows.push_back(static_cast<uint32_t>(strtoul(tok.c_str(), NULL, 10)));
        if (const char *f = getenv("ZDW_SPLIT_FIELDS")) {
            std::stringstream fs(f);
            while (std::getline(fs, tok, ',')) if (!tok.empty()) fields.push_back(static_cast<uint32_t>(strtoul(tok.c_str(), NULL, 10)));
        }
        setBlockSplits(rows, fields);
    }

    try {
        const bool several = devices.size() > 1 || (devices.size() == 1 && getenv("ZDW_MULTI_ALWAYS"));
        const bool whole = several || getenv("ZDW_CLI_WHOLE_INPUT");      // several devices cut a host image into their pieces
        if (bStreamingInput) {
            if (!whole) return processFile(NULL, NULL, 0, 0 /* stdin */, schema, filestub, bValidate, exeName, outputDir, zArgs, meta);
            std::vector<unsigned char> image;
            unsigned char buf[1 << 16];
            size_t got;
            while ((got = fread(buf, 1, sizeof buf, stdin)) > 0) image.insert(image.end(), buf, buf + got);
            return processFile(&image, NULL, image.size(), -1, schema, filestub, bValidate, exeName, outputDir, zArgs, meta);
        }
        const std::string sqlName = std::string(filestub) + "." + getInputFileExtension();
        const int fd = open(sqlName.c_str(), O_RDONLY);
        if (fd < 0) return MISSING_SQL_FILE;
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return MISSING_SQL_FILE; }
        const uint64_t len = static_cast<uint64_t>(st.st_size);
        if (!whole && (!S_ISREG(st.st_mode) || getenv("ZDW_CLI_READ_INPUT"))) {
            // not a regular file (a FIFO ...): read in pieces by the library while the blocks are encoded, length unknown
            const ERR_CODE r = processFile(NULL, NULL, S_ISREG(st.st_mode) ? len : 0, fd, schema, filestub, bValidate, exeName, outputDir, zArgs, meta);
            close(fd);
            return r;
        }
        // a regular file: mapped and copied to the device from the page cache (read() into a pinned buffer first costs a host
        // copy: 12 GB/s against 44-50)
        const char *map = NULL;
        if (len) {
            void *m = mmap(NULL, len, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); return MISSING_SQL_FILE; }
            map = static_cast<const char *>(m);
        }
        const ERR_CODE r = processFile(NULL, map, len, -1, schema, filestub, bValidate, exeName, outputDir, zArgs, meta);
        if (map) munmap(const_cast<char *>(map), len);
        close(fd);
        return r;
    } catch (const std::bad_alloc &) {
        return OUT_OF_MEMORY;
    }
}

}  // namespace zdw
}  // namespace adobe
