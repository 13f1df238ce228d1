"""GPU: the decode path (zdw_dec_*, zdw_amd/csrc/zdw_decode.hip) against the reference's own files and outputs.
The yardsticks: the committed sources of the reference's golden files, the Python test decoder (tests/zdw_decode.py, pinned on the CPU
by tests/test_decode_cpu.py) rendered with the reference's printing rules, and the shipped host decoder on larger inputs."""
import hashlib
import os
import subprocess

import pytest

from tests import parity_util as pu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sandbox():
    s = pu.CliSandbox()
    yield s
    s.close()


@pytest.fixture()
def dec():
    from zdw_amd import capi
    d = capi.Decoder(0)
    yield d
    d.close()


def test_reference_fixtures_decode_to_their_sources(dec):
    """test.zdw (v9), analytics-hits.zdw (v10, 2086 columns), movie_tickets.zdw: byte for byte the committed .sql -- what the
    reference's Scala tests check (format/.../ZDWStreamIteratorTest.scala:104-131)"""
    for zdw, sql in (("test.zdw", "test.sql"), ("analytics-hits.zdw.xz", "analytics-hits.sql.xz"), ("movie_tickets.zdw.xz", "movie_tickets.sql.xz")):
        rc, text, blocks = dec.decode_all(pu.rd(zdw))
        assert rc == 0, (zdw, dec.error())
        assert text == pu.rd(sql), zdw
        assert sum(b.num_rows for b in blocks) == pu.rd(sql).count(b"\n")


def test_reference_vectors_decode_like_the_test_decoder(dec):
    """every v11 stream the compiled reference produced (edge + fuzz vectors: escapes, char(1) pairs, signed and 64-bit values,
    unused columns, dictionary widths, long rows, metadata)"""
    from tests import zdw_decode
    from tests.test_decode_cpu import _render
    n = 0
    for fname in ("edge_cases.json", "fuzz_cases.json"):
        for c in pu.cases(fname):
            if c["exit_code"] != 0 or not c["zdw"]:
                continue
            rc, text, _ = dec.decode_all(c["zdw"])
            assert rc == 0, (c["name"], dec.error())
            z, rows = zdw_decode.decode(c["zdw"])
            assert text == _render(z, rows), c["name"]
            n += 1
    assert n >= 200


def test_multi_block_stream_and_errors(dec, sandbox):
    """three blocks (forced splits, a partial row between them) decode to the input's rows; a stream cut short says so (6); an
    unsupported version is refused (5)"""
    from tests import zdw_decode
    from tests.test_decode_cpu import _render
    from zdw_amd import synth
    desc, types, tsv = synth.make_table(seed=91, ncols=30, nrows=5000)
    rc, z = sandbox.run(desc, tsv, args=["-q"], env={"ZDW_SPLIT_ROWS": "1200,1700", "ZDW_SPLIT_FIELDS": "4,0"})
    assert rc == 0
    rc, text, blocks = dec.decode_all(z)
    assert rc == 0 and [b.num_rows for b in blocks] == [1200, 1700, 2100] and [b.is_last_block for b in blocks] == [0, 0, 1]
    zz, rows = zdw_decode.decode(z)
    assert text == _render(zz, rows)
    assert sum(b.zdw_bytes for b in blocks) + 0 <= len(z)
    rc, _, _ = dec.decode_all(z[:-5])
    assert rc == 6, dec.error()
    rc, _, _ = dec.decode_all(b"\x0c\x00" + z[2:])
    assert rc == 5
    # crafted dictionary headers: sizes that would wrap a position, or that the format cannot hold, are corrupt data (9), never a fault
    zz0 = zdw_decode.decode(z)[0]
    at = zz0.header_len + 9                      # the first block's dictionary section: index size, then the size field
    isz = z[at]
    assert 1 <= isz <= 4
    for bad in (bytes([8]) + b"\xf0" + b"\xff" * 7,            # size near 2^64: position + size wraps
                bytes([8]) + (1 << 33).to_bytes(8, "little"),   # beyond the format's 32-bit offsets
                bytes([9]) + b"\x01" * 9,                       # an index size no version writes
                bytes([isz]) + (len(z) * 2).to_bytes(8, "little")[:isz]):      # longer than the stream
        rc, _, _ = dec.decode_all(z[:at] + bad + z[at + 1 + isz:])
        assert rc == 9, (bad, rc, dec.error())


@pytest.mark.parametrize("ncols,group", [(70, ""), (130, ""), (190, "1"), (190, "2"), (300, ""), (500, ""), (500, "8"), (513, "")])
def test_column_groups_of_the_row_kernel(dec, sandbox, ncols, group, monkeypatch):
    """Schemas of 2 .. 8 batches of 64 columns (and one just beyond, which takes the direct kernel): a wave puts together a group of
    batches of a row, groups of one row meet at the offsets the lengths kernel left.  Against the Python test decoder."""
    from tests import zdw_decode
    from tests.test_decode_cpu import _render
    from zdw_amd import synth
    if group:
        monkeypatch.setenv("ZDW_DEC_GROUP", group)
    desc, types, tsv = synth.make_table(seed=200 + ncols, ncols=ncols, nrows=700)
    rc, z = sandbox.run(desc, tsv, args=["-q"])
    assert rc == 0
    rc, text, blocks = dec.decode_all(z)
    assert rc == 0, dec.error()
    zz, rows = zdw_decode.decode(z)
    want = _render(zz, rows)
    assert text == want, next(i for i in range(min(len(text), len(want))) if text[i] != want[i]) if len(text) == len(want) else (len(text), len(want))


def test_rows_that_all_repeat_the_row_above(dec, sandbox):
    """A table whose rows are all the same: after the first, every row is nothing but zero flag bytes, and on such data row chains
    of different phase never meet -- the guesses of the parallel walk are wrong and are corrected pass by pass.  The text is still
    exact.  (Also: a block without any used column.)"""
    row = b"abc\t17\tx\t\t2024-01-05 10:00:00\t-5\n"
    desc = b"a\tvarchar(20)\nb\tint(11) unsigned\nc\tchar(1)\nd\tvarchar(9)\ne\tdatetime\nf\tint(11)\n"
    n = 400000
    rc, z = sandbox.run(desc, row * n, args=["-q"])
    assert rc == 0
    rc, text, blocks = dec.decode_all(z)
    assert rc == 0 and blocks[0].num_rows == n
    assert text == row * n
    assert blocks[0].walk_passes >= 1
    rc, z = sandbox.run(desc, b"\t\t\t\t\t\n" * 1000, args=["-q"])
    assert rc == 0
    rc, text, blocks = dec.decode_all(z)
    assert rc == 0 and text == b"\t0\t\t\t\t0\n" * 1000


def test_large_block_against_the_host_decoder(dec, sandbox, tmp_path):
    """1 GB of the benchmark's table through the command line, then both decoders: the same text (sha256), which is the input with
    empty integer / decimal fields printed as 0 / 0.000000000000"""
    import torch
    from zdw_amd import build, synth
    table = synth.DeviceTable(ncols=200, device="cuda:0")
    rows = table.rows_for_bytes(1.0e9)
    buf, n = table.generate(0, rows)
    tsv = buf[:n].cpu().numpy().tobytes()
    del buf
    rc, z = sandbox.run(table.desc, tsv, args=["-q"])
    assert rc == 0
    rc, text, blocks = dec.decode_all(z)
    assert rc == 0 and sum(b.num_rows for b in blocks) == rows, dec.error()
    f = tmp_path / "big.zdw"
    f.write_bytes(z)
    p = subprocess.run([build.DECODER_EXE, "-q", "-", str(f)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 0
    assert len(text) == len(p.stdout) and hashlib.sha256(text).hexdigest() == hashlib.sha256(p.stdout).hexdigest()
    # and against the input itself, on its first rows
    integer = [t in (7, 8, 9, 10, 12, 13, 14, 15) for t in table.types]
    want = bytearray()
    for line in tsv.split(b"\n", 2000)[:2000]:
        fs = line.split(b"\t")
        want += b"\t".join((x or b"0") if integer[i] else (x or b"0.000000000000") if table.types[i] == 11 else x for i, x in enumerate(fs)) + b"\n"
    assert text[:len(want)] == bytes(want)
    print("decode stages (ms):", dec.stage_ms(), "walk passes", blocks[0].walk_passes, "text bytes", len(text))


def test_decoder_binary_gpu_mode(tmp_path):
    """unconvertDWfile --gpu: the same bytes on stdout and in the files as the host mode, the same exit code for a stream cut short"""
    from zdw_amd import build
    exe = build.DECODER_EXE
    for zdw, sql in (("test.zdw", "test.sql"), ("movie_tickets.zdw.xz", "movie_tickets.sql.xz")):
        p = subprocess.run([exe, "--gpu", "-q", "-", pu.G + "/" + zdw], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert p.returncode == 0, p.stderr
        assert p.stdout == pu.rd(sql), zdw
    p = subprocess.run([exe, "--gpu", "-o", "-", pu.G + "/test.zdw"], stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == pu.rd("test.desc.sql")
    (tmp_path / "t.zdw").write_bytes(pu.rd("test.zdw"))
    assert subprocess.run([exe, "--gpu", "-q", "-d", str(tmp_path), str(tmp_path / "t.zdw")]).returncode == 0
    assert (tmp_path / "t.sql").read_bytes() == pu.rd("test.sql") and (tmp_path / "t.desc.sql").read_bytes() == pu.rd("test.desc.sql")
    (tmp_path / "bad.zdw").write_bytes(pu.rd("test.zdw")[:-3])
    assert subprocess.run([exe, "--gpu", "-q", "-t", str(tmp_path / "bad.zdw")], stderr=subprocess.PIPE).returncode == 6
