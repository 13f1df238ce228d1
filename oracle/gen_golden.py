#!/usr/bin/env python3
"""Generate tests/golden/*.json by running the COMPILED REFERENCE (oracle/_ref/convertDWfile).

TEST INFRASTRUCTURE ONLY.  Needs /root/reference (to build oracle/_ref via `make -C oracle`);
the committed JSON fixtures are what travels.  Each case stores the inputs (schema text, TSV
bytes, CLI flags, optional .metadata sidecar) and what the reference produced (exit code, raw
.zdw bytes with the compressor child replaced by `cat`, and the `Row N had the problem` line).

The cases are the list in SURVEY.md §8(c): behaviours that the reference's own committed
fixtures (test-files/) do not pin.

    python oracle/gen_golden.py               # rewrites tests/golden/edge_cases.json, fuzz_cases.json
    python oracle/gen_golden.py --multiblock  # rewrites tests/golden/multiblock.json (two-block files, Q9)
"""
import base64
import json
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")


def b64(b):
    return base64.b64encode(b).decode() if b is not None else None


def desc_of(cols):
    """cols: list of (name, sqltype) -> .desc.sql text"""
    return "".join("%s\t%s\n" % (n, t) for n, t in cols).encode()


def rows(*rs):
    return b"".join(b"\t".join(f if isinstance(f, bytes) else f.encode() for f in r) + b"\n" for r in rs)


CASES = []


def case(name, desc, tsv, args=(), meta_file=None):
    CASES.append(dict(name=name, desc=desc, tsv=tsv, args=list(args), meta_file=meta_file))


S2 = [("a", "varchar(20)"), ("b", "varchar(20)")]
S3 = [("s", "varchar(50)"), ("n", "int(11)"), ("t", "varchar(50)")]

# (1) escapes ---------------------------------------------------------------------------------
case("esc_tab_in_field", desc_of(S2), b"x\\\ty\tz\n" b"p\tq\n")
case("esc_newline_in_field", desc_of(S2), b"line1\\\nline2\tz\n" b"p\tq\n")
case("double_backslash_before_tab", desc_of(S2), b"bs\\\\\tz\n" b"bs\\\\\\\tstill\tw\n")
case("field_ending_backslashes", desc_of(S2), b"a\\\\\\\\\tb\\\\\n" b"a\\\\\\\\\tb\\\\\n")
case("row_starts_with_tab", desc_of(S2), b"\tz\n" b"\t\n" b"q\t\n")
case("row_starts_with_backslashes", desc_of(S2), b"\\\\\tz\n" b"\\\ta\tz\n")
case("esc_newline_then_blank", desc_of(S2), b"a\\\n\n" b"\tb\n" b"c\td\n")
case("esc_newline_then_tab", desc_of(S2), b"a\\\n\tb\n" b"c\td\n")
case("esc_newline_chain", desc_of(S2), b"a\\\nb\\\nc\\\n\td\n" b"e\tf\n")
# (2) blank lines, tails ----------------------------------------------------------------------
case("blank_lines", desc_of(S2), b"\n\na\tb\n\n\nc\td\n\n")
case("one_byte_tail", desc_of(S2), b"a\tb\nc\td\nq")
case("unterminated_last_row", desc_of(S2), b"a\tb\nlast\t5")
case("unterminated_escaped_tail", desc_of(S2), b"a\tb\nlast\t5\\\n")
case("only_newlines", desc_of(S2), b"\n\n\n")
case("single_char_rows", desc_of([("a", "varchar(5)")]), b"x\ny\n\nx\n")
# (3) CRLF ------------------------------------------------------------------------------------
case("crlf", desc_of(S3), b"a\t4\tb\r\n" b"a\t4\r\tb\r\n" b"\r\t5\t\r\n")
# (4) integers --------------------------------------------------------------------------------
INT_TYPES = ["tinyint(3) unsigned", "tinyint(4)", "smallint(5) unsigned", "smallint(6)", "int(11) unsigned",
             "int(11)", "bigint(20) unsigned", "bigint(20)"]
case("ints_all_types", desc_of([("c%d" % i, t) for i, t in enumerate(INT_TYPES)]),
     rows(*[[str(v + i) for i in range(8)] for v in (5, 200, 70000, 5, 0, 4000000000, 1 << 40)]))
case("ints_negative", desc_of([("n", "int(11)"), ("m", "bigint(20)")]),
     rows(["-5", "7"], ["", "-1"], [" 12", "18446744073709551615"], ["12abc", "99999999999999999999999"], ["0", "0"]))
case("ints_syntax", desc_of([("n", "int(11) unsigned")]),
     rows(["7"], ["7"], ["+3"], ["0x10"], ["0"], ["-0"], ["+"], ["-"], [" \x0b\x0c\r9"], ["1e5"], ["00012"],
          ["18446744073709551616"], ["-18446744073709551616"], ["9223372036854775808"], ["-9223372036854775808"]))
case("ints_all_zero_col", desc_of([("z", "tinyint(3) unsigned"), ("n", "int(11)")]), rows(["0", "1"], ["0", "2"], ["", "2"]))
case("ints_width_steps", desc_of([("n%d" % i, "bigint(20) unsigned") for i in range(9)]),
     rows(["1"] * 9, [str((1 << (8 * i)) + 0) for i in range(8)] + ["18446744073709551615"],
          [str(1 << (8 * i)) if i else "256" for i in range(8)] + ["2"]))
case("ints_range_255_256", desc_of([("a", "int(11)"), ("b", "int(11)")]), rows(["1", "1"], ["255", "256"], ["3", "257"]))
case("int_col_first_seen_later", desc_of([("a", "int(11)"), ("b", "varchar(9)")]), rows(["", "x"], ["0", "x"], ["9", ""], ["9", "y"]))
# (5) CHAR ------------------------------------------------------------------------------------
case("char_plain", desc_of([("c", "char(1)"), ("s", "varchar(9)")]), rows(["x", "1"], ["y", "1"], ["", "2"], ["x", "2"]))
case("char_escaped", desc_of([("c", "char(1)"), ("s", "varchar(9)")]),
     b"x\t1\n" b"\\\t\t2\n" b"\t3\n" b"\\\\\t4\n" b"\\\n\t5\n")
case("char_high_bit", desc_of([("c", "char(1)")]), b"x\n\xe9\n\n\xe9\n\x80\n")
case("char_two_bytes", desc_of([("c", "char(1)"), ("d", "char(1)")]), rows(["ab", "q"], ["ac", "q"], ["a", "q\xff".encode("latin1")], ["a\x80".encode("latin1"), "r"]))
# (6) other dictionary types ------------------------------------------------------------------
case("char2_decimal_datetime", desc_of([("c2", "char(2)"), ("d", "decimal(24,12)"), ("dt", "datetime"), ("tx", "text"),
                                        ("tt", "tinytext"), ("mt", "mediumtext"), ("lt", "longtext"), ("c3", "char(3)")]),
     rows(["US", "1.500000000000", "2020-01-02 03:04:05", "t", "u", "v", "w", "abc"],
          ["DE", "-2.25", "2020-01-02 03:04:05", "t", "", "v", "w2", "abd"],
          ["US", "1.500000000000", "0000-00-00 00:00:00", "", "u", "US", "w", "abc"]))
# (7) degenerate files -------------------------------------------------------------------------
case("empty_file", desc_of(S3), b"")
case("all_empty_fields", desc_of(S3), b"\t0\t\n\t\t\n")
case("single_row", desc_of(S3), b"a\t1\tb\n")
case("single_column", desc_of([("n", "int(11)")]), b"5\n6\n6\n")
# (8) wrong column count ----------------------------------------------------------------------
case("wrong_cols_row3", desc_of(S3), b"a\t1\tb\n" b"a\t2\tb\n" b"a\t3\n" b"a\t4\tb\n")
case("wrong_cols_row1_extra", desc_of(S2), b"a\tb\tc\n")
case("wrong_cols_after_blank", desc_of(S2), b"a\tb\n\n\nc\n")
# (9) many columns ----------------------------------------------------------------------------
_c12 = [("c%d" % i, "int(11)") for i in range(12)]
case("twelve_int_cols", desc_of(_c12),
     rows(*[[("0" if i in (1, 5) else str(r * (i + 1) % 7 + 1)) for i in range(12)] for r in (1, 2, 3, 3, 4)]))
_c70 = [("v%d" % i, "varchar(9)" if i % 3 else "int(11)") for i in range(70)]
case("seventy_cols_sparse", desc_of(_c70),
     rows(*[[("" if (i * 7 + r) % 5 == 0 or i in (3, 8, 64) else ("w%d" % ((i + r) % 4) if i % 3 else str(r + i)))
             for i in range(70)] for r in range(6)]))
# (10) long rows ------------------------------------------------------------------------------
def long_row(total_with_nl):
    return b"k\t" + b"z" * (total_with_nl - 3) + b"\n"


for L in (16382, 16383, 16384, 16385, 32767, 32768):
    case("long_row_%d" % L, desc_of(S2), b"a\tb\n" + long_row(L) + b"a\tb\n")
case("long_row_esc_newline_boundary", desc_of(S2), b"k\t" + b"z" * 16379 + b"\\\n" + b"tail\n" + b"a\tb\n")
case("long_row_esc_newline_short", desc_of(S2), b"k\t" + b"z" * 100 + b"\\\n" + b"y" * 17000 + b"\n" + b"a\tb\n")
# (11) dictionary size transitions ------------------------------------------------------------
def dict_of_size(total):
    """TSV (1 varchar column) whose dictionary getSize() == total (sum(len+1) + 1)."""
    need = total - 1
    out, i = [], 0
    while need > 0:
        ln = min(9, need - 1)
        if need - (ln + 1) == 1:   # would leave a remainder that cannot hold a 1-char string
            ln -= 1
        s = ("%09d" % i)[:ln] if ln > 0 else ""
        assert len(s) == ln and ln >= 1
        out.append(s)
        need -= ln + 1
        i += 1
    return rows(*[[s] for s in out])


for total in (255, 256, 257, 65535, 65536):
    case("dict_size_%d" % total, desc_of([("s", "varchar(20)"), ]), dict_of_size(total))
# (12) metadata -------------------------------------------------------------------------------
case("metadata_cli", desc_of(S2), b"a\tb\n", args=["--metadata:zeta=1", "--metadata:alpha=two=2", "--metadata:zeta=3"])
case("metadata_sidecar", desc_of(S2), b"a\tb\n", meta_file=b"k2=v2\n\nk1=v=1\n")
case("metadata_sidecar_bad", desc_of(S2), b"a\tb\n", meta_file=b"k2=v2\nnoequals\n")
case("metadata_empty_value", desc_of(S2), b"a\tb\n", args=["--metadata:k="])
# (13) schema variants ------------------------------------------------------------------------
case("desc_variants", (b"Field\tType\tNull\n" b"a\tchar(3)\n" b"b\tmediumint(8) unsigned\n" b"c\tfloat unsigned\n" b"d\tdouble\n"
                       b"e\tdate\n" b"f\ttimestamp\n" b"g\tvarchar(70000)\n" b"h\t decimal(5,2)\n" b"i\tDATETIME\n"
                       b"j\tint(10) unsigned zerofill\tNO\n" b"k\tcharacter\n" b"l\tchar(2)\n" b"m\ttinytext\n"),
     rows(["abc", "70000", "3.5", "3.99", "2020-01-02", "2020-01-02 00:00:00", "g", "1.25", "2021-05-06", "0042", "kk", "xy", "tt"],
          ["abd", "1", "1", "-3.5", "2021-01-02", "2020", "g", "1.25", "2021", "42", "kk", "xz", ""]))
case("desc_missing_tab", b"a\tint(11)\nbroken line\n", b"1\n")
# (14) trimming -------------------------------------------------------------------------------
case("trim_spaces", desc_of([("c", "char(1)"), ("s", "varchar(9)"), ("n", "int(11)")]),
     rows(["x  ", "ab  ", "5 "], ["  ", "ab", "5"], [" ", "  ", " "], ["x", " ab ", " 7  "]), args=["-t"])
case("no_trim_spaces", desc_of([("c", "char(1)"), ("s", "varchar(9)"), ("n", "int(11)")]),
     rows(["x  ", "ab  ", "5 "], ["  ", "ab", "5"], [" ", "  ", " "], ["x", " ab ", " 7  "]))
case("trim_backslash_space", desc_of(S2), b"a\\ \tb\n" b"a\\\\  \tb \n" b"\\ \t \n", args=["-t"])
# (16) shared / prefix / high bytes ------------------------------------------------------------
case("dict_order", desc_of(S2),
     rows([b"ab", b"abc"], [b"a", b"\xff"], [b"B", b"\xc3\xa9"], [b"Z", b"~"], [b"ab", b"ab"], [b"abc", b"a"], [b"\x7f", b"\x80"], [b"\x01", b"a\x01"]))
case("test_sql_copy", desc_of([("firstName", "varchar(100)"), ("lastName", "varchar(100)"), ("age", "int(11)"), ("eventCode", "int(11)")]),
     b"John\tSmith\t25\t1\nJane\tDoe\t21\t2\nJohn\tSmith\t25\t2\nJane\tDoe\t21\t1\n")


# ---------------------------------------------------------------------------------------------
# seeded fuzz cases: random schemas, random rows over a hostile token alphabet
# ---------------------------------------------------------------------------------------------
SQL_TYPES = ["varchar(255)", "varchar(20)", "text", "char(1)", "char(2)", "datetime", "decimal(24,12)",
             "tinyint(3) unsigned", "smallint(5)", "int(11)", "int(11) unsigned", "bigint(20) unsigned", "bigint(20)",
             "tinytext", "mediumtext", "longtext"]
TOK_TEXT = [b"a", b"b", b"ab", b"abc", b"0", b"7", b" ", b"  ", b"\\\\", b"\\\t", b"\\\n", b"\r", b"\xc3\xa9", b"\xff", b"\x01",
            b"~", b"Z", b"-", b"x", b"\\\\\\\\", b"http://", b"foo", b"bar"]
TOK_INT = [b"0", b"1", b"5", b"12", b"255", b"256", b"65535", b"70000", b"4294967296", b"-3", b"+9", b" 4", b"9x", b"",
           b"18446744073709551615", b"99999999999999999999", b"007", b"\r3"]


def fuzz_case(seed):
    rnd = random.Random(seed)
    ncols = rnd.choice([1, 2, 3, 5, 8, 9, 17, 40])
    cols = [("f%d" % i, rnd.choice(SQL_TYPES)) for i in range(ncols)]
    nrows = rnd.choice([1, 2, 5, 20, 60])
    pool = {}
    out = []
    bad_row = rnd.randrange(nrows) if rnd.random() < 0.08 else -1
    for rowi in range(nrows):
        if rnd.random() < 0.08:
            out.append(b"\n")
            continue
        fields = []
        for c, (_, t) in enumerate(cols):
            r = rnd.random()
            if r < 0.25:
                f = b""
            elif r < 0.55 and c in pool:
                f = pool[c]                      # repeat the previous row's value
            elif "int" in t:
                f = rnd.choice(TOK_INT)
            elif t == "char(1)":
                f = rnd.choice([b"x", b"y", b"\\\t", b"\\\\", b"\xe9", b"q", b"ab", b" "])
            else:
                f = b"".join(rnd.choice(TOK_TEXT) for _ in range(rnd.choice([1, 1, 2, 3, 6])))
            pool[c] = f
            fields.append(f)
        if rowi == bad_row:
            fields.append(b"extra")             # wrong column count
        out.append(b"\t".join(fields) + b"\n")
    tsv = b"".join(out)
    if rnd.random() < 0.15:
        tsv += b"tail-without-newline"
    args = ["-t"] if rnd.random() < 0.3 else []
    return dict(name="fuzz_%d" % seed, desc=desc_of(cols), tsv=tsv, args=args, meta_file=None)


def run(c):
    rc, zdw, so, se = oracle.run_ref(c["desc"], c["tsv"], args=c["args"], metadata_file=c["meta_file"])
    m = re.search(rb"Row (\d+) had the problem", so + se)
    code = re.search(rb"Internal error code=(\d+)", se)
    return dict(name=c["name"], desc=b64(c["desc"]), tsv=b64(c["tsv"]), args=c["args"], meta_file=b64(c["meta_file"]),
                exit_code=rc, zdw=b64(zdw), bad_row=int(m.group(1)) if m else None,
                err_code=int(code.group(1)) if code else None)


# (15) forced multi-block files (probe P19) -----------------------------------------------
# The inputs are too large to commit (>= 64 MiB of unique strings are needed before the reference's
# StringHeap allocates its second arena, stringheap.cpp:74-86): they are regenerated from the recipe
# below, and the fixture keeps what the reference produced for them: sha256 of the raw stream and the
# block summary read from the reference's OWN output (rows, dictionary entries, `done`, m_LongestLine).
# partial_fields = leading fields of the row after the block that pass 1 had visited when the limit
# tripped (they stay in the block's dictionary and ranges, ConvertToZDW.cpp:338,404): read off the
# entry count, since every string of these tables is unique.
MULTIBLOCK = [
    dict(name="p19_two_text_columns", mem_limit=100, nrows=1500000,
         desc=[("a", "varchar(64)"), ("b", "varchar(64)")], row=["A%042d", "B%042d"], text_cols=2),
    # an int column in front of the strings: the partial row's number pollutes the block's minimum
    # (values fall with the row number), so the column base of block 1 shows the pollution
    dict(name="partial_row_pollutes_min", mem_limit=215, nrows=1000000,
         desc=[("n", "int(11)"), ("s", "varchar(40)"), ("t", "varchar(60)")], row=["%(neg)d", "S%023d", "T%051d"], text_cols=2),
]


def multiblock_tsv(spec):
    out = []
    for i in range(spec["nrows"]):
        out.append("\t".join(f % {"neg": 2000000 - i} if "%(neg)" in f else f % i for f in spec["row"]))
    return ("\n".join(out) + "\n").encode()


def run_multiblock(spec):
    import hashlib
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
    import zdw_walk
    tsv = multiblock_tsv(spec)
    rc, zdw, so, se = oracle.run_ref(desc_of(spec["desc"]), tsv, args=["-q", "--mem-limit=%d" % spec["mem_limit"]])
    assert rc == 0 and zdw is not None, (rc, se[-300:])
    z = zdw_walk.walk(zdw)
    blocks = []
    for b in z.blocks:
        blocks.append(dict(rows=b.num_rows, entries=b.dict_blob.count(b"\0") - 1 if b.index_size else 0, done=b.done,
                           longest_line=b.longest_line, col_size=list(b.col_size), col_base=b.col_base))
    assert len(blocks) >= 2, "the reference did not split: lower mem_limit"
    # text fields of the partial row that were inserted = entries beyond the block's own rows
    extra = blocks[0]["entries"] - spec["text_cols"] * blocks[0]["rows"]
    ntext_before = [sum(1 for (_, t) in spec["desc"][:k] if "char" in t) for k in range(len(spec["desc"]) + 1)]
    partial = max(k for k in range(len(spec["desc"]) + 1) if ntext_before[k] == extra and (k == 0 or "char" in spec["desc"][k - 1][1]))
    return dict(name=spec["name"], mem_limit=spec["mem_limit"], nrows=spec["nrows"], desc=b64(desc_of(spec["desc"])), row=spec["row"],
                tsv_bytes=len(tsv), tsv_sha256=hashlib.sha256(tsv).hexdigest(), zdw_bytes=len(zdw), zdw_sha256=hashlib.sha256(zdw).hexdigest(),
                blocks=blocks, split_rows=[b["rows"] for b in blocks[:-1]], partial_fields=[partial] + [0] * (len(blocks) - 2))


def main():
    if "--multiblock" in sys.argv:
        if not oracle.have_ref():
            sys.exit("oracle/_ref/convertDWfile missing: run `make -C oracle` where /root/reference exists")
        res = [run_multiblock(s) for s in MULTIBLOCK]
        with open(os.path.join(OUT_DIR, "multiblock.json"), "w") as f:
            json.dump(dict(generator="oracle/gen_golden.py --multiblock", reference="adobe/zdw convertDWfile v11b (oracle/_ref), --mem-limit as listed",
                           cases=res), f, indent=1)
        for r in res:
            print(r["name"], r["zdw_sha256"][:16], [(b["rows"], b["entries"], b["done"]) for b in r["blocks"]], "partial", r["partial_fields"])
        return
    if not oracle.have_ref():
        sys.exit("oracle/_ref/convertDWfile missing: run `make -C oracle` where /root/reference exists")
    os.makedirs(OUT_DIR, exist_ok=True)
    edge = [run(c) for c in CASES]
    fuzz = [run(fuzz_case(s)) for s in range(1, 161)]
    for name, lst in (("edge_cases.json", edge), ("fuzz_cases.json", fuzz)):
        with open(os.path.join(OUT_DIR, name), "w") as f:
            json.dump(dict(generator="oracle/gen_golden.py", reference="adobe/zdw convertDWfile v11b (oracle/_ref)",
                           cases=lst), f, indent=0)
        print(name, len(lst), "cases", os.path.getsize(os.path.join(OUT_DIR, name)), "bytes")
    bad = [c["name"] for c in edge + fuzz if c["exit_code"] != 0]
    print("non-zero exit:", bad)


if __name__ == "__main__":
    main()
