"""Helpers shared by the GPU parity tests: run the HIP path (through the C ABI / the in-process
command line) and compare with expected bytes section by section."""
import base64
import json
import lzma
import os
import shutil
import stat
import tempfile

from tests import zdw_walk

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rd(name):
    p = os.path.join(G, name)
    return lzma.open(p).read() if name.endswith(".xz") else open(p, "rb").read()


def cases(fname):
    with open(os.path.join(G, fname)) as f:
        out = json.load(f)["cases"]
    for c in out:
        for k in ("desc", "tsv", "meta_file", "zdw"):
            c[k] = base64.b64decode(c[k]) if c[k] is not None else None
    return out


def multiblock_cases():
    """tests/golden/multiblock.json: two-block files the compiled reference wrote under --mem-limit
    (oracle/gen_golden.py --multiblock).  The inputs are regenerated from their recipe (multiblock_tsv)."""
    with open(os.path.join(G, "multiblock.json")) as f:
        out = json.load(f)["cases"]
    for c in out:
        c["desc"] = base64.b64decode(c["desc"])
    return out


def multiblock_tsv(c):
    import hashlib
    rows = []
    for i in range(c["nrows"]):
        rows.append("\t".join(f % {"neg": 2000000 - i} if "%(neg)" in f else f % i for f in c["row"]))
    tsv = ("\n".join(rows) + "\n").encode()
    assert len(tsv) == c["tsv_bytes"] and hashlib.sha256(tsv).hexdigest() == c["tsv_sha256"], "recipe drifted from the fixture"
    return tsv


class CliSandbox:
    """A scratch directory plus a PATH whose `gzip` is `cat`, so <x>.zdw.gz is the raw ZDW stream
    (the same trick oracle/_ref uses for the reference binary); ZDW_GZIP_THREADS=0 makes the command line use that child, as the
    reference does, instead of deflating in the process."""

    def __init__(self):
        self.dir = tempfile.mkdtemp(prefix="zdwgpu_")
        shim = os.path.join(self.dir, "shim")
        os.makedirs(shim)
        p = os.path.join(shim, "gzip")
        with open(p, "w") as f:
            f.write("#!/bin/sh\nexec cat\n")
        os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
        self.old_path = os.environ.get("PATH", "")
        os.environ["PATH"] = shim + os.pathsep + self.old_path
        self.old_threads = os.environ.get("ZDW_GZIP_THREADS")
        os.environ["ZDW_GZIP_THREADS"] = "0"          # the compressor is the child on PATH (the shim), not the writer in the process
        self.n = 0

    def close(self):
        os.environ["PATH"] = self.old_path
        if self.old_threads is None:
            os.environ.pop("ZDW_GZIP_THREADS", None)
        else:
            os.environ["ZDW_GZIP_THREADS"] = self.old_threads
        shutil.rmtree(self.dir, ignore_errors=True)

    def run(self, desc, tsv, args=(), meta_file=None, env=None):
        """Returns (exit_code, raw zdw bytes or None)."""
        from zdw_amd import cli
        self.n += 1
        d = os.path.join(self.dir, "c%d" % self.n)
        os.makedirs(d)
        stem = os.path.join(d, "t")
        with open(stem + ".sql", "wb") as f:
            f.write(tsv)
        with open(stem + ".desc.sql", "wb") as f:
            f.write(desc)
        if meta_file is not None:
            with open(stem + ".metadata", "wb") as f:
                f.write(meta_file)
        saved = {}
        for k, v in (env or {}).items():
            saved[k] = os.environ.get(k)
            os.environ[k] = v
        try:
            rc = cli.main_inprocess(list(args) + [stem + ".sql"])
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        out = stem + ".zdw.gz"
        data = open(out, "rb").read() if os.path.exists(out) else None
        shutil.rmtree(d, ignore_errors=True)
        return rc, data


def first_difference(got: bytes, want: bytes):
    """Human-readable location of the first differing byte, by section of the expected file."""
    if got == want:
        return None
    n = min(len(got), len(want))
    i = next((k for k in range(n) if got[k] != want[k]), n)
    where = "?"
    try:
        z = zdw_walk.walk(want)
        if i < z.header_len:
            where = "file header"
        for bi, b in enumerate(z.blocks):
            if b.start <= i < b.end:
                o = i - b.start
                if o < 9:
                    where = "block %d header" % bi
                elif o < 9 + len(b.dict_section):
                    where = "block %d dictionary (+%d of %d)" % (bi, o - 9, len(b.dict_section))
                elif o < 9 + len(b.dict_section) + len(b.stats_section):
                    where = "block %d column stats (+%d)" % (bi, o - 9 - len(b.dict_section))
                else:
                    ro = o - 9 - len(b.dict_section) - len(b.stats_section)
                    import bisect
                    r = bisect.bisect_right(b.row_offsets, ro) - 1
                    where = "block %d rows (+%d, row %d)" % (bi, ro, r)
    except Exception as e:  # noqa: BLE001
        where = "unwalkable expected (%s)" % e
    return "first difference at byte %d of %d/%d in %s: got %s want %s" % (
        i, len(got), len(want), where, got[i:i + 8].hex(), want[i:i + 8].hex())
