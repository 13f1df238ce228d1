// unconvertDWfile.cpp — host-side decoder for ZDW v9-v11 files: the sibling binary that
// `convertDWfile -v` pipes into `cmp` (cplusplus/ConvertToZDW.cpp:166-223), and a plain
// "<x>.zdw[.gz|.bz2|.xz|.zst] -> <x>.sql + <x>.desc.sql" tool.
//
// It follows the row loop of UnconvertFromZDW<T>::readNextRow (cplusplus/UnconvertFromZDW.cpp:1270-1462)
// for the column types the encoder writes: a column whose flag bit is clear repeats the previous row's
// value (0 before a block's first row); text-like values are dictionary offsets (0 = empty); char(1)
// holds one byte, or two after a backslash; unsigned / signed integers print value + base (a stored 0
// prints "0"); an unused column prints its default ("" for text, "0" for integers, "0.000000000000"
// for decimal: outputDefault, :1224-1267).  The reference's column selection (-c*), statistics, metadata
// output and pre-v9 formats are not provided: this is the validation path, not a replacement of the
// reference's decoder.  Plain C++; with --gpu the blocks are decoded on the device instead, through the zdw_dec_* calls of
// libzdw_hip.so (loaded from the directory of this binary when asked for: the host path needs no GPU library).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <dlfcn.h>
#include <unistd.h>

namespace {

enum ErrCode {   // cplusplus/zdw/UnconvertFromZDW.h:34-56 (values are API)
    OK = 0, BAD_PARAMETER = 1, GZREAD_FAILED = 2, FILE_CREATION_ERR = 3, FILE_OPEN_ERR = 4, UNSUPPORTED_ZDW_VERSION_ERR = 5,
    ZDW_LONGER_THAN_EXPECTED_ERR = 6, UNEXPECTED_DESC_TYPE = 7, ROW_COUNT_ERR = 8, CORRUPTED_DATA_ERROR = 9
};

enum : uint8_t {   // cplusplus/zdw_column_type_constants.h:17-35
    T_VARCHAR = 0, T_TEXT = 1, T_DATETIME = 2, T_CHAR_2 = 3, T_CHAR = 6, T_TINY = 7, T_SHORT = 8, T_LONG = 9, T_LONGLONG = 10,
    T_DECIMAL = 11, T_TINY_SIGNED = 12, T_SHORT_SIGNED = 13, T_LONG_SIGNED = 14, T_LONGLONG_SIGNED = 15, T_TINYTEXT = 16,
    T_MEDIUMTEXT = 17, T_LONGTEXT = 18
};

bool is_text(uint8_t t)
{
    return t == T_VARCHAR || t == T_TEXT || t == T_DATETIME || t == T_CHAR_2 || t == T_TINYTEXT || t == T_MEDIUMTEXT || t == T_LONGTEXT;
}
bool is_unsigned_int(uint8_t t) { return t == T_TINY || t == T_SHORT || t == T_LONG || t == T_LONGLONG; }
bool is_signed_int(uint8_t t) { return t == T_TINY_SIGNED || t == T_SHORT_SIGNED || t == T_LONG_SIGNED || t == T_LONGLONG_SIGNED; }

bool ends_with(const std::string &s, const char *suffix)
{
    const size_t n = strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

// the whole (decompressed) file in memory; compressed inputs come through the same external tools the encoder pipes into
int slurp(const std::string &path, std::vector<unsigned char> &out)
{
    const char *tool = nullptr;
    if (ends_with(path, ".gz")) tool = "gzip -dc";
    else if (ends_with(path, ".bz2")) tool = "bzip2 -dc";
    else if (ends_with(path, ".xz")) tool = "xz -dc";
    else if (ends_with(path, ".zst")) tool = "zstd -dc";
    FILE *f = nullptr;
    if (tool) {
        FILE *probe = fopen(path.c_str(), "rb");
        if (!probe) return FILE_OPEN_ERR;
        fclose(probe);
        std::string cmd = std::string(tool) + " '" + path + "'";
        f = popen(cmd.c_str(), "r");
    } else {
        f = fopen(path.c_str(), "rb");
    }
    if (!f) return FILE_OPEN_ERR;
    unsigned char buf[1 << 16];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) out.insert(out.end(), buf, buf + n);
    const int rc = tool ? pclose(f) : fclose(f);
    return rc == 0 ? OK : (tool ? GZREAD_FAILED : FILE_OPEN_ERR);
}

struct Reader {
    const unsigned char *p, *end;
    bool need(size_t n) const { return (size_t)(end - p) >= n; }
    uint64_t le(unsigned bytes)
    {
        uint64_t v = 0;
        for (unsigned i = 0; i < bytes; ++i) v |= (uint64_t)p[i] << (8 * i);
        p += bytes;
        return v;
    }
};

struct Out {
    FILE *f;
    std::vector<char> buf;
    explicit Out(FILE *file) : f(file) { buf.reserve(1 << 20); }
    void put(const void *s, size_t n)
    {
        if (buf.size() + n > (1u << 20)) flush();
        if (n > (1u << 20)) { fwrite(s, 1, n, f); return; }
        buf.insert(buf.end(), static_cast<const char *>(s), static_cast<const char *>(s) + n);
    }
    void put_u64(uint64_t v)   // llutoa, UnconvertFromZDW.cpp:318-330
    {
        char t[24]; size_t pos = sizeof t;
        do { t[--pos] = (char)('0' + v % 10); v /= 10; } while (v);
        put(t + pos, sizeof t - pos);
    }
    void put_i64(int64_t v)    // lltoa, :333-352
    {
        if (v < 0) { put("-", 1); put_u64((uint64_t)0 - (uint64_t)v); } else put_u64((uint64_t)v);
    }
    void flush() { if (!buf.empty()) { fwrite(buf.data(), 1, buf.size(), f); buf.clear(); } }
};

std::string column_desc(const std::string &name, uint8_t type, unsigned charsize)   // getColumnDesc, UnconvertFromZDW.cpp:625-667
{
    std::string s = name + "\t";
    char t[32];
    switch (type) {
    case T_VARCHAR: snprintf(t, sizeof t, "varchar(%u)", charsize ? charsize : 255u); s += t; break;
    case T_TEXT: s += "text"; break;
    case T_TINYTEXT: s += "tinytext"; break;
    case T_MEDIUMTEXT: s += "mediumtext"; break;
    case T_LONGTEXT: s += "longtext"; break;
    case T_DATETIME: s += "datetime"; break;
    case T_CHAR_2: s += "char(2)"; break;
    case T_CHAR: s += "char(1)"; break;
    case T_TINY: s += "tinyint(3) unsigned"; break;
    case T_SHORT: s += "smallint(5) unsigned"; break;
    case T_LONG: s += "int(11) unsigned"; break;
    case T_LONGLONG: s += "bigint(20) unsigned"; break;
    case T_TINY_SIGNED: s += "tinyint(3)"; break;
    case T_SHORT_SIGNED: s += "smallint(5)"; break;
    case T_LONG_SIGNED: s += "int(11)"; break;
    case T_LONGLONG_SIGNED: s += "bigint(20)"; break;
    case T_DECIMAL: s += "decimal(24,12)"; break;
    default: return std::string();
    }
    return s + "\n";
}

void put_default(Out &o, uint8_t t)   // outputDefault, UnconvertFromZDW.cpp:1224-1267
{
    if (is_unsigned_int(t) || is_signed_int(t)) o.put("0", 1);
    else if (t == T_DECIMAL) o.put("0.000000000000", 14);
    // text-like and char(1): empty
}

int decode(const std::vector<unsigned char> &data, Out *rows_out, FILE *desc_out, uint64_t *rows_total)
{
    Reader r{data.data(), data.data() + data.size()};
    if (!r.need(2)) return CORRUPTED_DATA_ERROR;
    const unsigned version = (unsigned)r.le(2);
    if (version < 9 || version > 11) return UNSUPPORTED_ZDW_VERSION_ERR;
    if (version >= 11) {                                        // metadata block (ConvertToZDW.cpp:677-694): skipped
        if (!r.need(4)) return CORRUPTED_DATA_ERROR;
        const uint64_t mlen = r.le(4);
        if (!r.need(mlen)) return CORRUPTED_DATA_ERROR;
        r.p += mlen;
    }
    std::vector<std::string> names;
    for (;;) {                                                  // name\0 ... then an extra \0 (:701-716)
        const unsigned char *z = static_cast<const unsigned char *>(memchr(r.p, 0, (size_t)(r.end - r.p)));
        if (!z) return CORRUPTED_DATA_ERROR;
        if (z == r.p) { ++r.p; break; }
        names.emplace_back(reinterpret_cast<const char *>(r.p), (size_t)(z - r.p));
        r.p = z + 1;
    }
    const size_t n = names.size();
    if (!r.need(3 * n)) return CORRUPTED_DATA_ERROR;
    std::vector<uint8_t> types(r.p, r.p + n);
    r.p += n;
    std::vector<unsigned> charsize(n);
    for (size_t c = 0; c < n; ++c) charsize[c] = (unsigned)r.le(2);
    for (size_t c = 0; c < n; ++c) {
        const std::string d = column_desc(names[c], types[c], charsize[c]);
        if (d.empty()) return UNEXPECTED_DESC_TYPE;
        if (desc_out) fwrite(d.data(), 1, d.size(), desc_out);
    }
    if (!rows_out) return OK;

    uint64_t total = 0;
    std::vector<uint64_t> val(n), base(n);
    std::vector<uint8_t> size(n);
    bool last = r.p == r.end;                                   // an empty input gives a header without blocks (:824-836)
    while (!last) {
        if (!r.need(9)) return CORRUPTED_DATA_ERROR;
        const uint64_t nrows = r.le(4);
        r.le(4);                                                // longest line: a buffer hint for the reference
        last = r.le(1) != 0;
        // dictionary (dictionary.cpp:76-111): 00, or index size, size, origin byte + strings
        if (!r.need(1)) return CORRUPTED_DATA_ERROR;
        const unsigned isz = (unsigned)r.le(1);
        const unsigned char *dict = nullptr;
        uint64_t dsize = 0;
        if (isz) {
            if (isz > 8 || !r.need(isz)) return CORRUPTED_DATA_ERROR;
            dsize = r.le(isz);
            if (!r.need(dsize) || dsize == 0 || r.p[dsize - 1] != 0) return CORRUPTED_DATA_ERROR;
            dict = r.p;
            r.p += dsize;
        }
        if (!r.need(n)) return CORRUPTED_DATA_ERROR;
        size_t used = 0;
        for (size_t c = 0; c < n; ++c) { size[c] = r.p[c]; if (size[c] > 8) return CORRUPTED_DATA_ERROR; used += size[c] != 0; }
        r.p += n;
        if (!r.need(8 * used)) return CORRUPTED_DATA_ERROR;
        for (size_t c = 0; c < n; ++c) { base[c] = size[c] ? r.le(8) : 0; val[c] = 0; }
        const size_t nflag = (used + 7) / 8;
        for (uint64_t row = 0; row < nrows; ++row) {
            if (!r.need(nflag)) return ZDW_LONGER_THAN_EXPECTED_ERR;
            const unsigned char *flags = r.p;
            r.p += nflag;
            size_t u = 0;
            for (size_t c = 0; c < n; ++c) {
                if (c) rows_out->put("\t", 1);
                const uint8_t t = types[c];
                if (!size[c]) { put_default(*rows_out, t); continue; }
                if (flags[u >> 3] >> (u & 7) & 1) {
                    if (!r.need(size[c])) return ZDW_LONGER_THAN_EXPECTED_ERR;
                    val[c] = r.le(size[c]);
                }
                ++u;
                const uint64_t v = val[c];
                if (is_text(t) || t == T_DECIMAL) {
                    if (!v) { put_default(*rows_out, t); continue; }
                    const uint64_t idx = v + base[c];
                    if (idx >= dsize) return CORRUPTED_DATA_ERROR;
                    const char *s = reinterpret_cast<const char *>(dict + idx);
                    rows_out->put(s, strlen(s));                // the blob ends with a NUL (checked above)
                } else if (t == T_CHAR) {                       // :1396-1420
                    if (!v) continue;
                    const uint64_t tuple = v + base[c];
                    char two[2] = {(char)tuple, (char)(tuple >> 8)};
                    if (two[0] != '\\') { if (two[0]) rows_out->put(two, 1); }
                    else rows_out->put(two, 2);
                } else if (is_unsigned_int(t)) {
                    rows_out->put_u64(v ? v + base[c] : 0);
                } else if (is_signed_int(t)) {
                    rows_out->put_i64((int64_t)(v ? v + base[c] : 0));
                } else {
                    return UNEXPECTED_DESC_TYPE;
                }
            }
            rows_out->put("\n", 1);
        }
        total += nrows;
    }
    if (r.p != r.end) return ZDW_LONGER_THAN_EXPECTED_ERR;
    if (rows_total) *rows_total = total;
    return OK;
}

// ---- --gpu: the same output through zdw_dec_* (include/zdw_hip.h) ----
struct GpuApi {
    void *lib = nullptr;
    int (*create)(int, void **) = nullptr;
    void (*destroy)(void *) = nullptr;
    const char *(*last_error)(const void *) = nullptr;
    int (*set_input_host)(void *, const unsigned char *, uint64_t) = nullptr;
    int (*header)(const void *, uint32_t *, uint32_t *) = nullptr;
    int (*column)(const void *, uint32_t, const char **, uint8_t *, uint16_t *) = nullptr;
    int (*next_block)(void *, void *) = nullptr;
    int (*text_write)(void *, int (*)(void *, const unsigned char *, uint64_t), void *, uint64_t *) = nullptr;
};
struct GpuBlock { uint32_t num_rows, longest_line, is_last_block, reserved; uint64_t text_bytes, zdw_bytes, dict_bytes, walk_passes; };      // zdw_dec_block

bool load_gpu(GpuApi &g, const char *argv0)
{
    std::string dir;
    char self[4096];
    const ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n > 0) { self[n] = 0; dir = self; const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? "." : dir.substr(0, sl); }
    else { dir = argv0; const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? "." : dir.substr(0, sl); }
    g.lib = dlopen((dir + "/libzdw_hip.so").c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!g.lib) g.lib = dlopen("libzdw_hip.so", RTLD_NOW | RTLD_LOCAL);
    if (!g.lib) { fprintf(stderr, "--gpu: %s\n", dlerror()); return false; }
    auto sym = [&](const char *name) { return dlsym(g.lib, name); };
    g.create = reinterpret_cast<decltype(g.create)>(sym("zdw_dec_create"));
    g.destroy = reinterpret_cast<decltype(g.destroy)>(sym("zdw_dec_destroy"));
    g.last_error = reinterpret_cast<decltype(g.last_error)>(sym("zdw_dec_last_error"));
    g.set_input_host = reinterpret_cast<decltype(g.set_input_host)>(sym("zdw_dec_set_input_host"));
    g.header = reinterpret_cast<decltype(g.header)>(sym("zdw_dec_header"));
    g.column = reinterpret_cast<decltype(g.column)>(sym("zdw_dec_column"));
    g.next_block = reinterpret_cast<decltype(g.next_block)>(sym("zdw_dec_next_block"));
    g.text_write = reinterpret_cast<decltype(g.text_write)>(sym("zdw_dec_text_write"));
    return g.create && g.destroy && g.last_error && g.set_input_host && g.header && g.column && g.next_block && g.text_write;
}

int write_rows(void *user, const unsigned char *bytes, uint64_t len)
{
    Out *o = static_cast<Out *>(user);
    o->flush();
    return fwrite(bytes, 1, len, o->f) == len ? 0 : 1;
}

int decode_gpu(const GpuApi &g, const std::vector<unsigned char> &data, Out *rows_out, FILE *desc_out, uint64_t *rows_total)
{
    void *d = nullptr;
    int rc = g.create(-1, &d);
    if (rc) { fprintf(stderr, "--gpu: %s\n", g.last_error(nullptr)); return rc == 20 ? FILE_OPEN_ERR : rc; }
    rc = g.set_input_host(d, data.data(), data.size());
    uint32_t version = 0, ncols = 0;
    if (!rc) rc = g.header(d, &version, &ncols);
    for (uint32_t c = 0; !rc && c < ncols; ++c) {
        const char *name = nullptr; uint8_t type = 0; uint16_t cs = 0;
        g.column(d, c, &name, &type, &cs);
        const std::string line = column_desc(name, type, cs);
        if (line.empty()) rc = UNEXPECTED_DESC_TYPE;
        else if (desc_out) fwrite(line.data(), 1, line.size(), desc_out);
    }
    uint64_t total = 0;
    while (!rc && rows_out) {
        GpuBlock b;
        rc = g.next_block(d, &b);
        if (rc) break;
        if (b.num_rows) { rc = g.text_write(d, write_rows, rows_out, nullptr); total += b.num_rows; }
        if (b.is_last_block) break;
    }
    if (rc) fprintf(stderr, "--gpu: %s\n", g.last_error(d));
    if (rows_total) *rows_total = total;
    g.destroy(d);
    return rc == 20 ? CORRUPTED_DATA_ERROR : rc;
}

void usage(const char *exe)
{
    printf("Usage: %s [-q] [-] [-o] [-t] [-w] [-d outputDirectory] file1 [file2...]\n"
           "\t-  write the rows to stdout (no .desc file, except with -o)\n"
           "\t-d <outputDirectory>  where to place the resulting files (default: next to the .zdw file)\n"
           "\t-o write the .desc file (to disk, or to stdout with -) and exit\n"
           "\t-q quiet\n"
           "\t-t test the integrity of the file only\n"
           "\t-w give outputted files no extension (default = .sql)\n"
           "\t--gpu decode the blocks on the GPU (libzdw_hip.so next to this binary)\n"
           "\t(column selection, statistics and metadata output of the reference tool are not provided)\n", exe);
}

}  // namespace

int main(int argc, char **argv)
{
    bool to_stdout = false, quiet = false, desc_only = false, test_only = false, no_ext = false, gpu = false;
    GpuApi api;
    std::string dir;
    std::vector<std::string> files;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-") to_stdout = true;
        else if (a == "-q") quiet = true;
        else if (a == "-o") desc_only = true;
        else if (a == "-t") test_only = true;
        else if (a == "-w") no_ext = true;
        else if (a == "-v") { }
        else if (a == "--gpu") gpu = true;
        else if (a == "-d") { if (++i >= argc) { usage(argv[0]); return BAD_PARAMETER; } dir = argv[i]; }
        else if (a == "--help") { usage(argv[0]); return OK; }
        else if (a == "--version") { printf("unconvertDWfile (zdw-mi355x) for ZDW v9-v11\n"); return OK; }
        else if (!a.empty() && a[0] == '-') { fprintf(stderr, "%s: unsupported option %s\n    Run with --help for usage info.\n", argv[0], a.c_str()); return BAD_PARAMETER; }
        else files.push_back(a);
    }
    if (files.empty()) { usage(argv[0]); return BAD_PARAMETER; }
    if (gpu && !load_gpu(api, argv[0])) return FILE_OPEN_ERR;
    auto run = [&](const std::vector<unsigned char> &data, Out *o, FILE *desc, uint64_t *rows) {
        return gpu ? decode_gpu(api, data, o, desc, rows) : decode(data, o, desc, rows);
    };
    int worst = OK;
    for (const std::string &path : files) {
        std::vector<unsigned char> data;
        int rc = slurp(path, data);
        uint64_t rows = 0;
        if (rc == OK) {
            if (test_only) {
                struct Null : Out { Null() : Out(fopen("/dev/null", "wb")) {} ~Null() { if (f) fclose(f); } } sink;
                rc = sink.f ? run(data, &sink, nullptr, &rows) : FILE_CREATION_ERR;
            } else if (to_stdout) {
                Out o(stdout);
                rc = run(data, desc_only ? nullptr : &o, desc_only ? stdout : nullptr, &rows);
                o.flush();
                fflush(stdout);
            } else {
                // <dir|same dir>/<basename up to ".zdw">[.sql] and .desc.sql
                std::string stem = path;
                const size_t z = stem.rfind(".zdw");
                if (z != std::string::npos) stem.resize(z);
                if (!dir.empty()) { const size_t sl = stem.rfind('/'); stem = dir + "/" + (sl == std::string::npos ? stem : stem.substr(sl + 1)); }
                FILE *fd = fopen((stem + ".desc" + (no_ext ? "" : ".sql")).c_str(), "wb");
                FILE *fr = desc_only ? nullptr : fopen((stem + (no_ext ? "" : ".sql")).c_str(), "wb");
                if (!fd || (!desc_only && !fr)) rc = FILE_CREATION_ERR;
                else {
                    Out o(fr ? fr : stdout);
                    rc = run(data, fr ? &o : nullptr, fd, &rows);
                    o.flush();
                }
                if (fd) fclose(fd);
                if (fr) fclose(fr);
            }
        }
        if (rc != OK) fprintf(stderr, "Error code=%d: %s: %s failed\n\n", rc, argv[0], path.c_str());
        else if (!quiet && !to_stdout) printf("%s: %llu rows\n", path.c_str(), (unsigned long long)rows);
        if (rc != OK) worst = rc;
    }
    return worst;
}
