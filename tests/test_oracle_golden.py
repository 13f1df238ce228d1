"""CPU tests: the oracle (oracle/zdw_oracle.c) against the reference's golden vectors.

(a) the reference's own committed fixtures (copied as data into tests/golden/): test.zdw (v9),
    analytics-hits.zdw(.xz) and movie_tickets.zdw.xz (v10).  The current encoder writes v11 =
    the same bytes with version word 0b 00 plus a 4-byte empty metadata length, so the check is
    oracle[6:] == golden[2:]  (SURVEY.md Q14).
(b) vectors produced by the compiled reference (oracle/gen_golden.py -> edge_cases.json, fuzz_cases.json).
"""
import base64
import hashlib
import json
import lzma
import os

import pytest

from oracle import oracle
from tests import zdw_walk

G = os.path.join(os.path.dirname(__file__), "golden")


def rd(name):
    p = os.path.join(G, name)
    if name.endswith(".xz"):
        return lzma.open(p).read()
    return open(p, "rb").read()


SHA_V11 = {  # sha256 of the raw v11 stream the compiled reference writes
    "test": "89304bb8898b5a184bc842d80ae7014fea5759556237c41c5671a7ec6eeb06ca",
    "movie_tickets": "081fb28a5c62817eee28bf5037ce52a5a3789f460a962b34bb8b32ef492080d7",
    "analytics-hits": "f810684347db8ae9f1315984ec58f955fc3bc963269913d8aae0ee18aa94327c",
}


@pytest.mark.parametrize("name,sql,zdw", [
    ("test", "test.sql", "test.zdw"),
    ("movie_tickets", "movie_tickets.sql.xz", "movie_tickets.zdw.xz"),
    ("analytics-hits", "analytics-hits.sql.xz", "analytics-hits.zdw.xz"),
])
def test_reference_committed_fixtures(name, sql, zdw):
    rc, out, info = oracle.encode(rd(name + ".desc.sql"), rd(sql))
    gold = rd(zdw)
    assert rc == 0
    assert out[:6] == b"\x0b\x00\x00\x00\x00\x00"
    assert gold[:2] in (b"\x09\x00", b"\x0a\x00")
    assert out[6:] == gold[2:]
    assert hashlib.sha256(out).hexdigest() == SHA_V11[name]
    z = zdw_walk.walk(out)
    assert len(z.blocks) == 1 and z.blocks[0].num_rows == info.rows_total


def _cases(fname):
    with open(os.path.join(G, fname)) as f:
        return json.load(f)["cases"]


def _args_to_kw(args):
    kw = dict(trim=False, metadata={})
    for a in args:
        if a == "-t":
            kw["trim"] = True
        elif a.startswith("--metadata:"):
            k, v = a[len("--metadata:"):].split("=", 1)
            kw["metadata"][k] = v
        else:
            raise AssertionError(a)
    return kw


def load_sidecar(blob):
    """ConvertToZDW::loadMetadataFile (ConvertToZDW.cpp:240-262): returns (bad_line or 0, dict)."""
    md = {}
    for i, line in enumerate(blob.split(b"\n"), 1):
        if i == blob.count(b"\n") + 1 and line == b"":
            break
        if not line:
            continue
        if b"=" not in line:
            return i, md
        k, v = line.split(b"=", 1)
        md[k] = v
    return 0, md


@pytest.mark.parametrize("fname", ["edge_cases.json", "fuzz_cases.json"])
def test_reference_generated_vectors(fname):
    n_ok = n_err = 0
    for c in _cases(fname):
        desc = base64.b64decode(c["desc"])
        tsv = base64.b64decode(c["tsv"])
        kw = _args_to_kw(c["args"])
        if not kw["metadata"] and c["meta_file"] is not None:
            bad, md = load_sidecar(base64.b64decode(c["meta_file"]))
            if bad:
                assert c["exit_code"] == 2 and c["err_code"] == 22, c["name"]
                n_err += 1
                continue
            kw["metadata"] = md
        rc, out, info = oracle.encode(desc, tsv, **kw)
        if c["exit_code"] == 0:
            assert rc == 0, c["name"]
            assert out == base64.b64decode(c["zdw"]), c["name"]
            zdw_walk.walk(out)
            n_ok += 1
        else:
            assert c["exit_code"] == 2, c["name"]       # the CLI collapses every failure to 2
            assert rc == c["err_code"], c["name"]
            if rc == 15:
                assert info.bad_row == c["bad_row"], c["name"]
            n_err += 1
    assert n_ok > 40 and n_err >= 1


@pytest.mark.parametrize("case", __import__("tests.parity_util", fromlist=["x"]).multiblock_cases(), ids=lambda c: c["name"])
def test_reference_multiblock_files(case):
    """Q9, pinned: two-block files the REAL reference wrote under --mem-limit (sha256 + block summary in
    tests/golden/multiblock.json).  Given the split the reference took (rows of block 1, fields of the partial
    row it had visited), the oracle's emulation must reproduce the reference's bytes."""
    from tests import parity_util as pu
    tsv = pu.multiblock_tsv(case)
    rc, out, info = oracle.encode(case["desc"], tsv, split_rows=case["split_rows"], split_cols=case["partial_fields"])
    assert rc == 0 and info.nblocks == len(case["blocks"])
    assert len(out) == case["zdw_bytes"]
    assert hashlib.sha256(out).hexdigest() == case["zdw_sha256"]
    z = zdw_walk.walk(out)
    for b, want in zip(z.blocks, case["blocks"]):
        assert (b.num_rows, b.done, b.longest_line) == (want["rows"], want["done"], want["longest_line"])
        assert list(b.col_size) == want["col_size"] and b.col_base == want["col_base"]
    # without the partial-row pollution the bytes differ: the fixture really pins it
    rc2, out2, _ = oracle.encode(case["desc"], tsv, split_rows=case["split_rows"], split_cols=[0] * len(case["split_rows"]))
    assert rc2 == 0 and hashlib.sha256(out2).hexdigest() != case["zdw_sha256"]


@pytest.mark.parametrize("case", __import__("tests.parity_util", fromlist=["x"]).multiblock_cases(), ids=lambda c: c["name"])
def test_memory_limit_model_finds_the_reference_split(case):
    """--mem-limit as a deterministic model (zdw_oracle.c: 2.6 + 64 x arenas + 64 B x dictionary entries, checked when the
    string heap takes a new 64 MiB arena): told only the limit, the oracle ends its blocks where the real reference did and
    writes the same bytes; one MB above the largest limit that trips, it writes one block like the reference."""
    from tests import parity_util as pu
    tsv = pu.multiblock_tsv(case)
    rc, out, info = oracle.encode(case["desc"], tsv, mem_limit_mb=case["mem_limit"])
    assert rc == 0 and info.nblocks == len(case["blocks"])
    assert hashlib.sha256(out).hexdigest() == case["zdw_sha256"]
    if case["name"] == "p19_two_text_columns":      # bisected with the reference build: two blocks up to 223, one from 224 on; none at all up to 66
        assert oracle.encode(case["desc"], tsv, mem_limit_mb=223)[2].nblocks == 2
        assert oracle.encode(case["desc"], tsv, mem_limit_mb=224)[2].nblocks == 1
        assert oracle.encode(case["desc"], tsv[:44000], mem_limit_mb=66)[0] == 7
        assert oracle.encode(case["desc"], tsv[:44000], mem_limit_mb=67)[0] == 0
