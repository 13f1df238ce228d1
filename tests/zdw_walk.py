"""Structural walker for raw (uncompressed) .zdw streams, v9-v11.  Test helper.

Splits a file into header / per-block sections and asserts that it is consumed to exact EOF.
Layout follows SURVEY.md §8(a) rows D-G (reference: cplusplus/ConvertToZDW.cpp:673-737,839-842,
dictionary.cpp:76-111, ConvertToZDW.cpp:475-479,588-589).
"""
import struct

TEXT_TYPES = {0, 1, 2, 3, 11, 16, 17, 18}


class Block:
    __slots__ = ("num_rows", "longest_line", "done", "index_size", "dict_size", "dict_blob", "col_size", "col_base",
                 "used", "rows", "row_offsets", "start", "end", "dict_section", "stats_section")


class ZdwFile:
    def __init__(self):
        self.version = None
        self.metadata = b""
        self.names = []
        self.types = b""
        self.char_sizes = []
        self.header_len = 0
        self.blocks = []


def walk(data: bytes, decode_rows=True) -> ZdwFile:
    z = ZdwFile()
    p = 0
    (z.version,) = struct.unpack_from("<H", data, p); p += 2
    if z.version >= 11:
        (mlen,) = struct.unpack_from("<I", data, p); p += 4
        z.metadata = data[p:p + mlen]; p += mlen
    while True:
        q = data.index(b"\0", p)
        if q == p:
            p += 1
            break
        z.names.append(data[p:q]); p = q + 1
    n = len(z.names)
    z.types = data[p:p + n]; p += n
    z.char_sizes = list(struct.unpack_from("<%dH" % n, data, p)); p += 2 * n
    z.header_len = p
    while p < len(data):
        b = Block()
        b.start = p
        b.num_rows, b.longest_line, b.done = struct.unpack_from("<IIB", data, p); p += 9
        d0 = p
        b.index_size = data[p]; p += 1
        if b.index_size == 0:
            b.dict_size = 0
            b.dict_blob = b""
        else:
            b.dict_size = int.from_bytes(data[p:p + b.index_size], "little"); p += b.index_size
            b.dict_blob = data[p:p + b.dict_size]; p += b.dict_size
            assert b.dict_blob[0] == 0
        b.dict_section = data[d0:p]
        s0 = p
        b.col_size = data[p:p + n]; p += n
        b.used = [c for c in range(n) if b.col_size[c]]
        b.col_base = list(struct.unpack_from("<%dQ" % len(b.used), data, p)); p += 8 * len(b.used)
        b.stats_section = data[s0:p]
        nflag = (len(b.used) + 7) // 8
        r0 = p
        b.row_offsets = []
        if decode_rows:
            sizes = [b.col_size[c] for c in b.used]
            for _ in range(b.num_rows):
                b.row_offsets.append(p - r0)
                flags = data[p:p + nflag]; p += nflag
                for u, sz in enumerate(sizes):
                    if flags[u >> 3] >> (u & 7) & 1:
                        p += sz
            b.rows = data[r0:p]
        else:
            assert b.done == 1, "cannot skip rows of a non-final block without decoding"
            b.rows = data[r0:]
            p = len(data)
        b.end = p
        z.blocks.append(b)
        if b.done:
            break
    assert p == len(data), "trailing bytes after last block: %d != %d" % (p, len(data))
    return z
