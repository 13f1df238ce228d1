"""A small ZDW (v9-v11) decoder and an independent TSV field interpreter, for round-trip property tests.

decode(zdw)            -> rows of values as the FILE stores them (text columns: bytes; numeric columns: u64)
interpret(types, tsv)  -> rows of values as the SOURCE implies them, following SURVEY.md Q1-Q7 directly
                          (unescaped separators, blank lines skipped, unterminated tail dropped,
                          strtoull for integers, the pass-2 char(1) value)
A correct encode satisfies decode(encode(tsv)) == interpret(tsv), whatever the block boundaries are.
Numeric zero and an empty numeric field are the same stored value (the format cannot tell them apart).
Pure Python: for small inputs only.
"""
from tests import zdw_walk

TEXT = {0, 1, 2, 3, 11, 16, 17, 18}
CHAR = 6
M64 = (1 << 64) - 1


def decode(data: bytes):
    z = zdw_walk.walk(data)
    out = []
    for b in z.blocks:
        used = b.used
        sizes = [b.col_size[c] for c in used]
        prev = [0] * len(used)
        nflag = (len(used) + 7) // 8
        p = 0
        rows = b.rows
        for _ in range(b.num_rows):
            flags = rows[p:p + nflag]
            p += nflag
            vals = list(prev)
            for u, sz in enumerate(sizes):
                if flags[u >> 3] >> (u & 7) & 1:
                    vals[u] = int.from_bytes(rows[p:p + sz], "little")
                    p += sz
            prev = vals
            row = []
            for c in range(len(z.types)):
                t = z.types[c]
                if b.col_size[c] == 0:
                    row.append(b"" if t in TEXT else 0)
                    continue
                u = used.index(c)
                v = vals[u]
                if t in TEXT:
                    if v == 0:
                        row.append(b"")
                    else:
                        e = b.dict_blob.index(b"\0", v)
                        row.append(b.dict_blob[v:e])
                else:
                    # stored values are truncated to the column width: compare modulo 2^(8*size)
                    row.append(("num", v, b.col_base[u], sizes[u]))
            out.append(row)
        assert p == len(rows)
    return z, out


def strtoull(s: bytes) -> int:
    i = 0
    while i < len(s) and (s[i] == 0x20 or 9 <= s[i] <= 13):
        i += 1
    neg = False
    if i < len(s) and s[i] in b"+-":
        neg = s[i] == 0x2d
        i += 1
    j = i
    while j < len(s) and 0x30 <= s[j] <= 0x39:
        j += 1
    if j == i:
        return 0
    v = int(s[i:j])
    if v > M64:
        return M64
    return (-v) & M64 if neg else v


def split_rows(tsv: bytes):
    """logical rows (terminator excluded): a newline ends a row iff an even number of backslashes precedes it"""
    rows, start, i, n = [], 0, 0, len(tsv)
    while i < n:
        if tsv[i] == 0x0a:
            k = 0
            while i - 1 - k >= start and tsv[i - 1 - k] == 0x5c:
                k += 1
            if k % 2 == 0:
                if i > start:
                    rows.append(tsv[start:i])
                start = i + 1
        i += 1
    return rows          # an unterminated tail is dropped


def split_fields(row: bytes):
    out, start = [], 0
    for i, ch in enumerate(row):
        if ch == 0x09:
            k = 0
            while i - 1 - k >= 0 and row[i - 1 - k] == 0x5c:
                k += 1
            if k % 2 == 0:
                out.append(row[start:i])
                start = i + 1
    out.append(row[start:])
    return out


def s8(b):
    return b - 256 if b >= 128 else b


def interpret(types, tsv: bytes, trim=False):
    rows = []
    for r in split_rows(tsv):
        f = split_fields(r)
        assert len(f) == len(types)
        row = []
        for t, x in zip(types, f):
            if trim:
                x = x.rstrip(b" ")
            if t in TEXT:
                row.append(x)
            elif t == CHAR:
                row.append(((s8(x[0]) + 256 * (s8(x[1]) if len(x) > 1 else 0)) & M64) if x else 0)
            else:
                row.append(strtoull(x) if x else 0)
        rows.append(row)
    return rows


def same(decoded_row, source_row):
    for d, s in zip(decoded_row, source_row):
        if isinstance(d, tuple):
            _, stored, base, size = d
            want = ((s - base) & M64) if s else 0
            if stored != want & ((1 << (8 * size)) - 1):
                return False
        elif isinstance(s, int):
            if s != 0 or d not in (0, b""):      # unused numeric column: every value was 0 / empty
                return False
        elif d != s:
            return False
    return True
