"""CPU: the several-thread gzip writer behind `.zdw.gz` (zdw_amd/csrc/parallel_gzip.h; the reference pipes into a `gzip` child,
ConvertToZDW.cpp:817-835).  Its file must be an ordinary single-member .gz holding the bytes it was given."""
import ctypes
import gzip
import os
import subprocess
import zlib

import numpy as np
import pytest

from zdw_amd import build


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(build.LIB_CLI):
        pytest.skip("libzdw_cli.so is not built")
    l = ctypes.CDLL(build.LIB_CLI)
    l.zdw_cli_gzip_bytes.restype = ctypes.c_int
    l.zdw_cli_gzip_bytes.argtypes = [ctypes.c_char_p, ctypes.c_ulonglong, ctypes.c_char_p, ctypes.c_uint, ctypes.c_int, ctypes.c_ulonglong]
    return l


def _payloads():
    rng = np.random.default_rng(5)
    words = [bytes(rng.integers(97, 123, size=int(rng.integers(1, 12)), dtype=np.uint8)) for _ in range(500)]
    text = b"\t".join(words[int(i)] for i in rng.integers(0, 500, size=1_400_000))          # ~9 MB, compressible: several pieces
    return [("empty", b""), ("one_byte", b"\x00"), ("below_a_piece", text[:70_000]), ("exactly_a_piece", text[:2 << 20]),
            ("a_piece_and_one", text[:(2 << 20) + 1]), ("several_pieces", text),
            ("incompressible", rng.integers(0, 256, size=5_000_000, dtype=np.uint8).tobytes())]


@pytest.mark.parametrize("threads,level,piece", [(1, 6, 0), (4, 6, 1 << 16), (8, 1, 4096 + 7), (3, 9, 3_000_000)])
def test_file_is_one_gzip_member_holding_the_bytes(lib, tmp_path, threads, level, piece):
    for name, data in _payloads():
        path = str(tmp_path / (name + ".gz"))
        assert lib.zdw_cli_gzip_bytes(data, len(data), path.encode(), threads, level, piece) == 0
        raw = open(path, "rb").read()
        assert raw[:4] == b"\x1f\x8b\x08\x00"
        # one member: a single raw-deflate stream between the 10-byte header and the 8-byte trailer
        d = zlib.decompressobj(-15)
        back = d.decompress(raw[10:])
        assert d.eof and back == data, name
        assert d.unused_data == (zlib.crc32(data) & 0xFFFFFFFF).to_bytes(4, "little") + (len(data) & 0xFFFFFFFF).to_bytes(4, "little"), name
        assert gzip.decompress(raw) == data, name
    # and the gzip program agrees (the decoders read `.zdw.gz` through `gzip -dc`)
    p = subprocess.run(["gzip", "-dc", path], stdout=subprocess.PIPE)
    assert p.returncode == 0 and p.stdout == data
    assert subprocess.run(["gzip", "-t", path]).returncode == 0


def test_unwritable_path_is_an_error(lib, tmp_path):
    assert lib.zdw_cli_gzip_bytes(b"abc", 3, str(tmp_path / "no_such_dir" / "x.gz").encode(), 2, 6, 0) == 2
