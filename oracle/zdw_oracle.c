/*
 * zdw_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * A scalar CPU restatement of adobe/zdw's TSV -> ZDW encode path, written from the
 * behaviour of the reference (not copied from it).  It exists so that tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg have a checker for the
 * HIP path on machines where /root/reference does not exist.  Nothing under
 * zdw_amd/ may include, link, load or call this file.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
 * byte-for-byte against (a) the reference's own committed fixtures
 * (test-files/test.zdw, analytics-hits.zdw, movie_tickets.zdw.xz; v9/v10 files, so
 * compared modulo the 6-byte v11 header difference) and (b) vectors produced by the
 * compiled reference itself (oracle/_ref/convertDWfile, built by oracle/Makefile from
 * the sources where they lie under /root/reference) and committed under tests/golden/.
 *
 * Reference anchors (paths relative to /root/reference/):
 *   desc parsing ........ cplusplus/ConvertToZDW.cpp:91-162   (ReadDescFile)
 *   logical row reader .. cplusplus/getnextrow.cpp:26-84      (GetNextRow)
 *   field split / trim .. cplusplus/ConvertToZDW.cpp:265-323, 1048-1067
 *   pass 1 .............. cplusplus/ConvertToZDW.cpp:329-414   (parseInput)
 *   dictionary .......... cplusplus/dictionary.cpp:31-111, dictionary.h:30-57
 *   column stats ........ cplusplus/ConvertToZDW.cpp:417-483   (writeLookupColumnStats)
 *   pass 2 .............. cplusplus/ConvertToZDW.cpp:486-606   (writeBlockRows)
 *   file/block framing .. cplusplus/ConvertToZDW.cpp:673-737, 765-894
 *   type ids ............ cplusplus/zdw_column_type_constants.h:17-35
 *   error codes ......... cplusplus/ConvertToZDW.h:44-70
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <stdio.h>

/* ---- column type ids: zdw_column_type_constants.h:17-35 ---- */
enum {
    T_VARCHAR = 0, T_TEXT = 1, T_DATETIME = 2, T_CHAR_2 = 3, T_CHAR = 6,
    T_TINY = 7, T_SHORT = 8, T_LONG = 9, T_LONGLONG = 10, T_DECIMAL = 11,
    T_TINY_SIGNED = 12, T_SHORT_SIGNED = 13, T_LONG_SIGNED = 14, T_LONGLONG_SIGNED = 15,
    T_TINYTEXT = 16, T_MEDIUMTEXT = 17, T_LONGTEXT = 18
};

/* ---- error codes: ConvertToZDW.h:44-70 (values frozen) ---- */
enum {
    E_OK = 0, E_OUT_OF_MEMORY = 7, E_DESC_FILE_MISSING_TYPE_INFO = 14,
    E_WRONG_NUM_OF_COLUMNS_ON_A_ROW = 15, E_UNKNOWN_ERROR = 20, E_BAD_METADATA_PARAM = 21
};

#define ZDWO_VERSION 11           /* ConvertToZDW.cpp:75 */
#define ZDWO_FIRST_ROW_CAP 16384u /* ConvertToZDW.cpp:965 */

typedef struct {
    uint8_t *p;
    size_t len, cap;
} obuf;

static int ob_reserve(obuf *b, size_t extra)
{
    if (b->len + extra <= b->cap) return 0;
    size_t nc = b->cap ? b->cap : 4096;
    while (nc < b->len + extra) nc *= 2;
    uint8_t *np = (uint8_t *)realloc(b->p, nc);
    if (!np) return -1;
    b->p = np; b->cap = nc;
    return 0;
}
static int ob_put(obuf *b, const void *src, size_t n)
{
    if (ob_reserve(b, n)) return -1;
    if (n) memcpy(b->p + b->len, src, n);
    b->len += n;
    return 0;
}
static int ob_le(obuf *b, uint64_t v, unsigned nbytes)
{
    uint8_t t[8];
    for (unsigned i = 0; i < 8; ++i) t[i] = (uint8_t)(v >> (8 * i));
    return ob_put(b, t, nbytes);
}

static int type_is_text(uint8_t t)
{
    switch (t) {
    case T_VARCHAR: case T_TEXT: case T_TINYTEXT: case T_MEDIUMTEXT: case T_LONGTEXT:
    case T_DATETIME: case T_CHAR_2: case T_DECIMAL: return 1;
    default: return 0;
    }
}
static int type_is_int(uint8_t t)
{
    return (t >= T_TINY && t <= T_LONGLONG) || (t >= T_TINY_SIGNED && t <= T_LONGLONG_SIGNED);
}

/* =====================================================================================
 * Schema (.desc.sql) parsing — ConvertToZDW.cpp:91-162
 * ===================================================================================== */
typedef struct {
    uint32_t ncols;
    uint8_t *types;     /* ncols */
    int32_t *charsize;  /* ncols (int in the reference; stored as u16 in the file) */
    char **names;       /* ncols, NUL-terminated */
} zdwo_schema;

void zdwo_schema_free(zdwo_schema *s)
{
    if (!s) return;
    for (uint32_t i = 0; i < s->ncols; ++i) free(s->names[i]);
    free(s->names); free(s->types); free(s->charsize);
    memset(s, 0, sizeof *s);
}

static int has_prefix(const char *s, const char *pre) { return strncmp(s, pre, strlen(pre)) == 0; }

/* Returns 0, or E_DESC_FILE_MISSING_TYPE_INFO when a line carries no tab. */
int zdwo_parse_desc(const char *text, size_t len, zdwo_schema *out)
{
    memset(out, 0, sizeof *out);
    size_t capc = 0, pos = 0;
    char line[1024];                       /* the reference reads with a 1024-byte fgets buffer (:93,103) */
    while (pos < len) {
        size_t n = 0;
        while (pos < len && n < sizeof line - 1) {
            char ch = text[pos++];
            line[n++] = ch;
            if (ch == '\n') break;
        }
        line[n] = 0;
        if (strncasecmp(line, "Field", 5) == 0) continue;          /* header line (:105) */
        char *tab = strchr(line, '\t');
        if (!tab) { zdwo_schema_free(out); return E_DESC_FILE_MISSING_TYPE_INFO; }
        *tab++ = 0;
        if (out->ncols == capc) {
            capc = capc ? capc * 2 : 64;
            out->types = (uint8_t *)realloc(out->types, capc);
            out->charsize = (int32_t *)realloc(out->charsize, capc * sizeof(int32_t));
            out->names = (char **)realloc(out->names, capc * sizeof(char *));
        }
        uint32_t c = out->ncols++;
        out->names[c] = strdup(line);
        out->charsize[c] = 0;
        uint8_t ty;
        if (has_prefix(tab, "varchar")) { ty = T_VARCHAR; out->charsize[c] = atoi(tab + 8); }
        else if (has_prefix(tab, "char")) {
            int cs = atoi(tab + 5);
            ty = cs == 1 ? T_CHAR : cs == 2 ? T_CHAR_2 : T_VARCHAR;
            out->charsize[c] = cs;
        }
        else if (has_prefix(tab, "text")) ty = T_TEXT;
        else if (has_prefix(tab, "tinytext")) ty = T_TINYTEXT;
        else if (has_prefix(tab, "mediumtext")) ty = T_MEDIUMTEXT;
        else if (has_prefix(tab, "longtext")) ty = T_LONGTEXT;
        else if (has_prefix(tab, "datetime")) ty = T_DATETIME;
        else if (has_prefix(tab, "decimal") || (tab[0] && has_prefix(tab + 1, "decimal"))) ty = T_DECIMAL;
        else {
            int is_signed = strstr(tab, "unsigned") == NULL;
            if (has_prefix(tab, "tinyint")) ty = is_signed ? T_TINY_SIGNED : T_TINY;
            else if (has_prefix(tab, "smallint")) ty = is_signed ? T_SHORT_SIGNED : T_SHORT;
            else if (has_prefix(tab, "bigint")) ty = is_signed ? T_LONGLONG_SIGNED : T_LONGLONG;
            else ty = is_signed ? T_LONG_SIGNED : T_LONG;
        }
        out->types[c] = ty;
    }
    return E_OK;
}

/* =====================================================================================
 * Logical row reader over a memory image — getnextrow.cpp:26-84
 *
 * The reference reads with fgets() into a buffer whose capacity (m_LongestLine) doubles
 * whenever a continuation read finds it exactly full; that capacity is written into
 * every block header, so the chunking is reproduced here, not just the row boundaries.
 * ===================================================================================== */
typedef struct {
    const uint8_t *buf;
    size_t n;        /* bytes in the image */
    size_t pos;      /* next unread byte   */
    uint32_t cap;    /* emulated row-buffer capacity, persists across blocks */
} rowreader;

/* bytes one fgets(.., size, ..) call would deliver starting at p (size-1 max, stops after '\n') */
static size_t chunk_len(const rowreader *r, size_t p, size_t size)
{
    size_t maxc = size - 1, m = 0;
    while (m < maxc && p + m < r->n) {
        uint8_t ch = r->buf[p + m++];
        if (ch == '\n') break;
    }
    return m;
}

static int ends_row(const uint8_t *row, size_t len)
{
    if (row[len - 1] != '\n') return 0;
    size_t k = 0;                              /* backslashes directly before the newline */
    while (k + 2 <= len && row[len - 2 - k] == '\\') ++k;
    return (k & 1) == 0;
}

/* Returns 1 and [*s,*e) (terminating newline excluded) or 0 at end of input. */
static int next_row(rowreader *r, size_t *s, size_t *e)
{
    for (;;) {
        if (r->pos >= r->n) return 0;
        size_t start = r->pos;
        size_t len = chunk_len(r, start, r->cap);
        if (len < 2) { r->pos = start + len; continue; }        /* blank line / 1-byte tail (:38-43) */
        while (!ends_row(r->buf + start, len)) {
            if (len == (size_t)r->cap - 1) r->cap *= 2;             /* buffer exactly full (:57-65) */
            size_t q = start + len;
            if (q >= r->n) { r->pos = r->n; return 0; }            /* unterminated tail is dropped (:67-69) */
            len += chunk_len(r, q, (size_t)r->cap - len);
        }
        r->pos = start + len;
        *s = start; *e = start + len - 1;
        return 1;
    }
}

/* =====================================================================================
 * Field split — ConvertToZDW.cpp:265-323 and 1048-1067
 * A tab separates iff an even number of backslashes directly precedes it (the run is
 * bounded by the start of the row).  With trimming, trailing 0x20 bytes are removed
 * from every field after the boundaries have been fixed.
 * ===================================================================================== */
typedef struct { size_t s, e; } span;

typedef struct {
    span *f;
    size_t n, cap;
} fieldvec;

static void split_row(const uint8_t *buf, size_t s, size_t e, int trim, fieldvec *fv)
{
    fv->n = 0;
    size_t fs = s;
    for (size_t i = s; i <= e; ++i) {
        int sep = 0;
        if (i == e) sep = 1;
        else if (buf[i] == '\t') {
            size_t k = 0;
            while (i - k > s && buf[i - 1 - k] == '\\') ++k;
            sep = (k & 1) == 0;
        }
        if (!sep) continue;
        if (fv->n == fv->cap) {
            fv->cap = fv->cap ? fv->cap * 2 : 256;
            fv->f = (span *)realloc(fv->f, fv->cap * sizeof(span));
        }
        size_t fe = i;
        if (trim) while (fe > fs && buf[fe - 1] == ' ') --fe;
        fv->f[fv->n].s = fs; fv->f[fv->n].e = fe; fv->n++;
        fs = i + 1;
    }
}

/* C-locale strtoull(s, NULL, 10) on the byte span [s,e) — ISO C semantics */
static uint64_t parse_u64(const uint8_t *p, size_t len)
{
    size_t i = 0;
    while (i < len && (p[i] == ' ' || (p[i] >= 9 && p[i] <= 13))) ++i;
    int neg = 0;
    if (i < len && (p[i] == '+' || p[i] == '-')) { neg = p[i] == '-'; ++i; }
    uint64_t v = 0; int any = 0, ovf = 0;
    for (; i < len; ++i) {
        unsigned d = (unsigned)p[i] - '0';
        if (d > 9) break;
        any = 1;
        if (v > UINT64_MAX / 10 || (v == UINT64_MAX / 10 && d > UINT64_MAX % 10)) ovf = 1;
        else v = v * 10 + d;
    }
    if (!any) return 0;
    if (ovf) return UINT64_MAX;
    return neg ? (uint64_t)0 - v : v;
}

/* char(1) value: pass 1 (:358-361) adds the 2nd byte only after a backslash, pass 2 (:543-547) always */
static uint64_t char_val(const uint8_t *p, size_t len, int pass2, uint64_t base)
{
    uint64_t v = (uint64_t)(int64_t)(int8_t)p[0];
    uint64_t hi = (uint64_t)(int64_t)((int)(int8_t)(len >= 2 ? p[1] : 0) * 256);
    if (pass2) { if (v) { v += hi; v -= base; } }
    else if (p[0] == '\\') v += hi;
    return v;
}

/* =====================================================================================
 * Dictionary — dictionary.cpp:31-111.  A set of unique byte strings; written in
 * unsigned-byte lexicographic order (shorter prefix first); offsets start at 1.
 * ===================================================================================== */
typedef struct {
    const uint8_t *p;
    uint32_t len;
    uint32_t off;
} dent;

typedef struct {
    dent *ent;  size_t n, cap;
    int64_t *slot; size_t nslot;  /* open addressing: index into ent or -1 */
    uint64_t bytes;               /* sum(len+1) */
} dict;

static uint64_t hash_bytes(const uint8_t *p, size_t len)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < len; ++i) { h ^= p[i]; h *= 1099511628211ull; }
    return h ^ (h >> 29);
}

static void dict_clear(dict *d) { d->n = 0; d->bytes = 0; for (size_t i = 0; i < d->nslot; ++i) d->slot[i] = -1; }

static void dict_rehash(dict *d, size_t nslot)
{
    free(d->slot);
    d->slot = (int64_t *)malloc(nslot * sizeof(int64_t));
    d->nslot = nslot;
    for (size_t i = 0; i < nslot; ++i) d->slot[i] = -1;
    for (size_t k = 0; k < d->n; ++k) {
        size_t i = hash_bytes(d->ent[k].p, d->ent[k].len) & (nslot - 1);
        while (d->slot[i] >= 0) i = (i + 1) & (nslot - 1);
        d->slot[i] = (int64_t)k;
    }
}

static int64_t dict_find(const dict *d, const uint8_t *p, size_t len)
{
    if (!d->nslot) return -1;
    size_t i = hash_bytes(p, len) & (d->nslot - 1);
    while (d->slot[i] >= 0) {
        const dent *e = &d->ent[d->slot[i]];
        if (e->len == len && memcmp(e->p, p, len) == 0) return d->slot[i];
        i = (i + 1) & (d->nslot - 1);
    }
    return -1;
}

/* returns 1 when the string is new to the dictionary */
static int dict_insert(dict *d, const uint8_t *p, size_t len)
{
    if (dict_find(d, p, len) >= 0) return 0;
    if ((d->n + 1) * 2 > d->nslot) dict_rehash(d, d->nslot ? d->nslot * 2 : 1024);
    if (d->n == d->cap) { d->cap = d->cap ? d->cap * 2 : 1024; d->ent = (dent *)realloc(d->ent, d->cap * sizeof(dent)); }
    d->ent[d->n].p = p; d->ent[d->n].len = (uint32_t)len; d->ent[d->n].off = 0;
    size_t i = hash_bytes(p, len) & (d->nslot - 1);
    while (d->slot[i] >= 0) i = (i + 1) & (d->nslot - 1);
    d->slot[i] = (int64_t)d->n++;
    d->bytes += len + 1;
    return 1;
}

static int dent_cmp(const void *a, const void *b)
{
    const dent *x = (const dent *)a, *y = (const dent *)b;
    size_t m = x->len < y->len ? x->len : y->len;
    int c = memcmp(x->p, y->p, m);
    if (c) return c;
    return x->len < y->len ? -1 : x->len > y->len;
}

static unsigned bytes_for(uint64_t v)   /* width rule shared by dictionary.cpp:62-73 and ConvertToZDW.cpp:458-464 */
{
    unsigned n = 1;
    while (v >= 256) { ++n; v /= 256; }
    return n;
}

/* writes the dictionary section and assigns offsets; returns the offset width */
static unsigned dict_write(dict *d, obuf *o)
{
    uint32_t total = (uint32_t)(d->bytes + 1);         /* getSize(): includes the origin byte; u32 like the reference */
    unsigned w = bytes_for(total);
    if (d->n == 0) { ob_le(o, 0, 1); return w; }
    qsort(d->ent, d->n, sizeof(dent), dent_cmp);
    ob_le(o, w, 1);
    ob_le(o, total, w);
    ob_le(o, 0, 1);
    uint32_t off = 1;
    for (size_t k = 0; k < d->n; ++k) {
        d->ent[k].off = off;
        ob_put(o, d->ent[k].p, d->ent[k].len);
        ob_le(o, 0, 1);
        off += d->ent[k].len + 1;
    }
    dict_rehash(d, d->nslot);                            /* entry indices moved with the sort */
    return w;
}

/* =====================================================================================
 * The encoder
 * ===================================================================================== */
typedef struct {
    uint32_t nblocks;
    uint32_t rows_total;
    uint32_t bad_row;           /* 1-based row within its block when rc == 15 */
    uint32_t longest_line;      /* final emulated m_LongestLine */
    uint64_t dict_entries;      /* last block */
    uint64_t dict_bytes;        /* last block, getSize() */
    uint32_t used_cols;         /* last block */
    uint32_t reserved;
    uint64_t header_bytes;
} zdwo_info;

static int meta_cmp(const void *a, const void *b)
{
    const char *const *x = (const char *const *)a, *const *y = (const char *const *)b;
    return strcmp(x[0], y[0]);   /* std::map<std::string,...> order == unsigned byte order for NUL-free keys */
}

/*
 * The reference's block limit (Q9), as a deterministic model.  A block ends when StringHeap allocates an arena
 * (64 MiB, or the string's size when that is more; the rest of the arena before it is wasted: stringheap.cpp:31-64)
 * while the process is at or beyond --mem-limit MB of virtual size (stringheap.cpp:74-86, memory.cpp:76-81,
 * default 3072: convertDWfile.cpp:161-163).  The virtual size is not reproducible across machines; what is
 * measured with the static reference build of oracle/Makefile on this image (bisecting --mem-limit on tables of
 * unique strings) fits
 *        vsize_MB = 2.6 + 64 * (arenas of this block) + 64 bytes * (strings in this block's std::map) / 2^20
 * to the megabyte: a red-black node of the map is a 64-byte malloc chunk, the arenas and the nodes go back to the
 * system when a block ends.  The string whose arena trips the limit is still inserted (dictionary.cpp:31-51); the
 * fields after it in its row are not visited and the row moves to the next block (ConvertToZDW.cpp:338,404).
 * 0 = no limit of this kind (blocks end only where split_rows says).
 */
static double g_mem_limit_mb = 0.0;
void zdwo_set_mem_limit(double mb) { g_mem_limit_mb = mb; }
#define ZDWO_ARENA_BYTES (64ull << 20)
#define ZDWO_VSIZE_BASE_MB 2.6
#define ZDWO_NODE_BYTES 64.0

/*
 * Encode a whole file image.
 *   meta_kv ....... 2*nmeta C strings (key0,value0,key1,value1,...), any order, later duplicates win
 *   split_rows .... forced block sizes: block i (all but the last) holds split_rows[i] rows
 *   split_cols .... how many leading fields of the following (uncounted) row were already
 *                   visited by pass 1 of block i when the memory limit tripped (Q9); may be NULL
 * The caller frees *out with zdwo_free().
 */
int zdwo_encode(const zdwo_schema *sc,
                const uint8_t *tsv, size_t tsv_len,
                const char *const *meta_kv, uint32_t nmeta,
                int trim,
                const uint32_t *split_rows, const uint32_t *split_cols, uint32_t nsplit,
                uint8_t **out, size_t *out_len, zdwo_info *info)
{
    obuf o = {0};
    zdwo_info inf; memset(&inf, 0, sizeof inf);
    int rc = E_OK;
    const uint32_t nc = sc->ncols;

    /* ---- metadata validation (ConvertToZDW.cpp:226-237) and header (:673-737) ---- */
    const char **kv = NULL; uint32_t nkv = 0;
    if (nmeta) {
        kv = (const char **)malloc(sizeof(char *) * 2 * nmeta);
        for (uint32_t i = 0; i < nmeta; ++i) {
            uint32_t j;
            for (j = 0; j < nkv; ++j) if (strcmp(kv[2 * j], meta_kv[2 * i]) == 0) break;
            kv[2 * j] = meta_kv[2 * i]; kv[2 * j + 1] = meta_kv[2 * i + 1];
            if (j == nkv) ++nkv;
        }
        qsort(kv, nkv, 2 * sizeof(char *), meta_cmp);
        for (uint32_t i = 0; i < nkv; ++i)
            if (strpbrk(kv[2 * i], "=\n") || strchr(kv[2 * i + 1], '\n')) { free(kv); return E_BAD_METADATA_PARAM; }
    }
    ob_le(&o, ZDWO_VERSION, 2);
    uint32_t mlen = 0;
    for (uint32_t i = 0; i < nkv; ++i) mlen += (uint32_t)(strlen(kv[2 * i]) + strlen(kv[2 * i + 1]) + 2);
    ob_le(&o, mlen, 4);
    for (uint32_t i = 0; i < nkv; ++i) {
        ob_put(&o, kv[2 * i], strlen(kv[2 * i]) + 1);
        ob_put(&o, kv[2 * i + 1], strlen(kv[2 * i + 1]) + 1);
    }
    free(kv);
    for (uint32_t c = 0; c < nc; ++c) ob_put(&o, sc->names[c], strlen(sc->names[c]) + 1);
    ob_le(&o, 0, 1);
    ob_put(&o, sc->types, nc);
    for (uint32_t c = 0; c < nc; ++c) ob_le(&o, (uint16_t)sc->charsize[c], 2);
    inf.header_bytes = o.len;

    /* ---- per-column state ---- */
    uint8_t *seen = (uint8_t *)calloc(nc ? nc : 1, 1);
    uint8_t *width = (uint8_t *)calloc(nc ? nc : 1, 1);
    uint64_t *cmin = (uint64_t *)calloc(nc ? nc : 1, 8), *cmax = (uint64_t *)calloc(nc ? nc : 1, 8);
    uint64_t *prev = (uint64_t *)calloc(nc ? nc : 1, 8);
    uint32_t *used = (uint32_t *)calloc(nc ? nc : 1, 4);
    uint8_t *rowbytes = (uint8_t *)malloc((size_t)nc * 8 + 8), *flags = (uint8_t *)malloc(nc / 8 + 2);
    fieldvec fv = {0};
    dict d; memset(&d, 0, sizeof d);
    rowreader rr = { tsv, tsv_len, 0, ZDWO_FIRST_ROW_CAP };

    uint32_t blk = 0;
    int more = 1;
    while (more) {
        /* ---------------- pass 1 (parseInput, :329-414) ---------------- */
        memset(seen, 0, nc);
        dict_clear(&d);
        uint32_t nrows = 0;
        const int forced = blk < nsplit;
        const size_t block_start = rr.pos;
        size_t s, e;
        int tripped = 0;
        uint64_t arena_free = 0, arenas = 0;       /* the memory-limit model: see g_mem_limit_mb */
        while (!tripped) {
            size_t save = rr.pos;
            if (!next_row(&rr, &s, &e)) break;
            split_row(tsv, s, e, trim, &fv);
            if (fv.n != nc) { rc = E_WRONG_NUM_OF_COLUMNS_ON_A_ROW; inf.bad_row = nrows + 1; goto done; }
            size_t limit = nc;
            if (forced && nrows == split_rows[blk]) {
                /* the row on which the reference ran out of memory: its first split_cols fields were
                 * visited (and stay in this block's dictionary / ranges) but the row is not counted */
                limit = split_cols ? split_cols[blk] : 0;
                if (limit > nc) limit = nc;
                tripped = 1;
                rr.pos = save;                      /* pass 2 re-reads exactly nrows rows; next block restarts here */
            }
            for (size_t c = 0; c < limit; ++c) {
                const uint8_t *p = tsv + fv.f[c].s; size_t len = fv.f[c].e - fv.f[c].s;
                if (!len) continue;                 /* empty fields are skipped entirely (:340-341) */
                uint8_t ty = sc->types[c];
                if (type_is_text(ty)) {
                    seen[c] = 1;
                    if (dict_insert(&d, p, len) && g_mem_limit_mb > 0.0) {
                        const uint64_t need = (uint64_t)len + 1;
                        if (need > arena_free) {               /* a new arena; what was left of the old one is wasted */
                            ++arenas;
                            arena_free = need > ZDWO_ARENA_BYTES ? need : ZDWO_ARENA_BYTES;
                            const double vsize = ZDWO_VSIZE_BASE_MB + 64.0 * (double)arenas + ZDWO_NODE_BYTES * (double)d.n / 1048576.0;
                            if (vsize >= g_mem_limit_mb && !tripped) {
                                tripped = 1;                   /* this row moves to the next block; fields after c stay unvisited */
                                rr.pos = save;
                                limit = c + 1;
                            }
                        }
                        arena_free -= need;
                    }
                    continue;
                }
                uint64_t v;
                if (ty == T_CHAR) v = char_val(p, len, 0, 0);
                else if (type_is_int(ty)) v = parse_u64(p, len);
                else continue;
                if (v > 0) {
                    if (seen[c]) { if (v > cmax[c]) cmax[c] = v; else if (v < cmin[c]) cmin[c] = v; }
                    else { cmin[c] = cmax[c] = v; seen[c] = 1; }
                }
            }
            if (!tripped) ++nrows;
        }
        const int last = !tripped;
        if (nrows == 0) {
            if (last) break;                       /* empty input: header only, rc OK (:824-836) */
            rc = E_OUT_OF_MEMORY; goto done;
        }
        /* ---------------- block header (:839-842) ---------------- */
        ob_le(&o, nrows, 4);
        ob_le(&o, rr.cap, 4);
        ob_le(&o, last ? 1 : 0, 1);
        /* ---------------- dictionary + column stats (:848-851) ---------------- */
        unsigned idxw = dict_write(&d, &o);
        uint32_t nused = 0;
        for (uint32_t c = 0; c < nc; ++c) {
            if (!seen[c]) { width[c] = 0; continue; }
            if (type_is_text(sc->types[c])) { width[c] = (uint8_t)idxw; cmin[c] = 0; }
            else { cmin[c] -= 1; width[c] = (uint8_t)bytes_for(cmax[c] - cmin[c]); }
            used[nused++] = c;
        }
        ob_put(&o, width, nc);
        for (uint32_t u = 0; u < nused; ++u) ob_le(&o, cmin[used[u]], 8);
        /* ---------------- pass 2 (writeBlockRows, :486-606) ---------------- */
        rowreader r2 = rr; r2.pos = block_start;   /* rewind; capacity no longer changes on a re-read */
        const size_t nflag = (nused + 7) / 8;
        memset(prev, 0, (size_t)nc * 8);
        for (uint32_t r = 0; r < nrows; ++r) {
            if (!next_row(&r2, &s, &e)) break;
            split_row(tsv, s, e, trim, &fv);
            memset(flags, 0, nflag);
            size_t nb = 0;
            for (uint32_t u = 0; u < nused; ++u) {
                uint32_t c = used[u];
                const uint8_t *p = tsv + fv.f[c].s; size_t len = fv.f[c].e - fv.f[c].s;
                uint8_t ty = sc->types[c];
                uint64_t v = 0;
                if (type_is_text(ty)) { if (len) v = d.ent[dict_find(&d, p, len)].off; }
                else if (ty == T_CHAR) { if (len) v = char_val(p, len, 1, cmin[c]); }
                else { v = len ? parse_u64(p, len) : 0; if (v > 0) v -= cmin[c]; }
                if (v != prev[c]) {
                    flags[u >> 3] |= (uint8_t)(1u << (u & 7));
                    for (unsigned j = 0; j < width[c]; ++j) rowbytes[nb++] = (uint8_t)(v >> (8 * j));
                }
                prev[c] = v;
            }
            ob_put(&o, flags, nflag);
            ob_put(&o, rowbytes, nb);
        }
        inf.rows_total += nrows;
        inf.dict_entries = d.n; inf.dict_bytes = d.bytes + 1; inf.used_cols = nused;
        ++blk; inf.nblocks = blk;
        more = !last;
    }
done:
    inf.longest_line = rr.cap;
    free(seen); free(width); free(cmin); free(cmax); free(prev); free(used); free(rowbytes); free(flags);
    free(fv.f); free(d.ent); free(d.slot);
    if (info) *info = inf;
    if (rc != E_OK) { free(o.p); *out = NULL; *out_len = 0; return rc; }
    *out = o.p; *out_len = o.len;
    return E_OK;
}

void zdwo_free(void *p) { free(p); }

/* convenience for ctypes: desc text in, everything else as above */
int zdwo_encode_desc(const char *desc, size_t desc_len,
                     const uint8_t *tsv, size_t tsv_len,
                     const char *const *meta_kv, uint32_t nmeta, int trim,
                     const uint32_t *split_rows, const uint32_t *split_cols, uint32_t nsplit,
                     uint8_t **out, size_t *out_len, zdwo_info *info)
{
    zdwo_schema sc;
    int rc = zdwo_parse_desc(desc, desc_len, &sc);
    if (rc) { *out = NULL; *out_len = 0; if (info) memset(info, 0, sizeof *info); return rc; }
    rc = zdwo_encode(&sc, tsv, tsv_len, meta_kv, nmeta, trim, split_rows, split_cols, nsplit, out, out_len, info);
    zdwo_schema_free(&sc);
    return rc;
}

/* schema accessors for ctypes */
uint32_t zdwo_schema_ncols(const zdwo_schema *s) { return s->ncols; }
const uint8_t *zdwo_schema_types(const zdwo_schema *s) { return s->types; }
const int32_t *zdwo_schema_charsizes(const zdwo_schema *s) { return s->charsize; }
const char *zdwo_schema_name(const zdwo_schema *s, uint32_t c) { return s->names[c]; }
