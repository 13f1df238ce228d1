"""CPU: the test decoder (tests/zdw_decode.py) read against the reference's own outputs -- every vector the
compiled reference encoded decodes to what its source says; this is the independent yardstick the GPU
round-trip tests use."""
from oracle import oracle
from tests import parity_util as pu
from tests import zdw_decode
from zdw_amd import synth


def test_reference_outputs_decode_to_their_sources():
    n = 0
    for fname in ("edge_cases.json", "fuzz_cases.json"):
        for c in pu.cases(fname):
            if c["exit_code"] != 0 or not c["zdw"]:
                continue
            z, rows = zdw_decode.decode(c["zdw"])
            src = zdw_decode.interpret(list(z.types), c["tsv"], trim="-t" in c["args"])
            assert len(rows) == len(src), c["name"]
            assert all(zdw_decode.same(d, s) for d, s in zip(rows, src)), c["name"]
            n += 1
    assert n >= 200


def test_multi_block_oracle_output_decodes():
    desc, types, tsv = synth.make_table(seed=41, ncols=14, nrows=2500, vocab=3000)
    rc, out, info = oracle.encode(desc, tsv, split_rows=[700, 900], split_cols=[3, 0])
    assert rc == 0 and info.nblocks == 3
    z, rows = zdw_decode.decode(out)
    src = zdw_decode.interpret(types, tsv)
    assert len(rows) == len(src) == 2500 and all(zdw_decode.same(d, s) for d, s in zip(rows, src))


# ---- the shipped decoder binary (zdw_amd/lib/unconvertDWfile, host-only C++) --------------------------------
def _decoder():
    import os
    import pytest
    from zdw_amd import build
    if not os.path.exists(build.DECODER_EXE):
        pytest.skip("unconvertDWfile is not built")
    return build.DECODER_EXE


def _render(z, rows):
    """TSV text as the reference decoder prints decoded values (UnconvertFromZDW.cpp:1224-1267,1340-1455)."""
    unsigned, signed = {7, 8, 9, 10}, {12, 13, 14, 15}
    out = []
    for row in rows:
        f = []
        for t, v in zip(z.types, row):
            if isinstance(v, tuple):
                _, stored, base, size = v
                full = (stored + base) & zdw_decode.M64 if stored else 0
                if t == zdw_decode.CHAR:
                    if not stored:
                        f.append(b"")
                    else:
                        b0, b1 = full & 0xff, (full >> 8) & 0xff
                        f.append(bytes([b0, b1]) if b0 == 0x5c else (bytes([b0]) if b0 else b""))
                elif t in signed:
                    f.append(b"%d" % (full - (1 << 64) if full >> 63 else full))
                else:
                    f.append(b"%d" % full)
            elif t in zdw_decode.TEXT:
                f.append(b"0.000000000000" if (t == 11 and v == b"") else v)
            else:                                  # unused numeric / char column
                f.append(b"" if t == zdw_decode.CHAR else b"0")
        out.append(b"\t".join(f) + b"\n")
    return b"".join(out)


def test_decoder_binary_on_the_reference_fixtures(tmp_path):
    """The reference's own golden files (v9 / v10) decode to the committed sources byte for byte -- what the
    reference's Scala tests check (format/.../ZDWStreamIteratorTest.scala:104-131) and what `convertDWfile -v` relies on."""
    import subprocess
    exe = _decoder()
    for zdw, sql in (("test.zdw", "test.sql"), ("analytics-hits.zdw.xz", "analytics-hits.sql.xz"), ("movie_tickets.zdw.xz", "movie_tickets.sql.xz")):
        p = subprocess.run([exe, "-q", "-", pu.G + "/" + zdw], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr
        assert p.stdout == pu.rd(sql), zdw
    p = subprocess.run([exe, "-o", "-", pu.G + "/test.zdw"], stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == pu.rd("test.desc.sql")
    (tmp_path / "bad.zdw").write_bytes(pu.rd("test.zdw")[:-3])
    assert subprocess.run([exe, "-q", "-t", str(tmp_path / "bad.zdw")], stderr=subprocess.PIPE).returncode == 6     # ZDW_LONGER_THAN_EXPECTED_ERR


def test_decoder_binary_on_the_reference_vectors(tmp_path):
    """Every v11 stream the compiled reference produced (edge + fuzz vectors), through the binary, against the
    Python test decoder rendered with the reference's printing rules."""
    import subprocess
    exe = _decoder()
    n = 0
    for fname in ("edge_cases.json", "fuzz_cases.json"):
        for c in pu.cases(fname):
            if c["exit_code"] != 0 or not c["zdw"]:
                continue
            f = tmp_path / "v.zdw"
            f.write_bytes(c["zdw"])
            p = subprocess.run([exe, "-q", "-", str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            assert p.returncode == 0, (c["name"], p.stderr)
            z, rows = zdw_decode.decode(c["zdw"])
            assert p.stdout == _render(z, rows), c["name"]
            n += 1
    assert n >= 200
