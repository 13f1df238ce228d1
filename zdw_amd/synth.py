"""Synthetic TSV tables of the benchmark shape (BASELINE.json configs[3]):
mixed VARCHAR / integer / char(1) / datetime / decimal columns, strings drawn from one shared
vocabulary with a Zipf(1.1) rank distribution, 25 % empty fields, each field repeating the
previous row's value with p = 0.3, no escapes.

Two generators with the same schema and distributions:
  * make_table()        numpy, host memory, for tests (thousands of rows);
  * DeviceTable         a HIP kernel pair (zdw_amd/csrc/synth_gen.hip -> lib/libzdw_synth.so) that
                        writes the table straight into HBM for the benchmark (GBs in milliseconds).
Both are counter-based: field (row, col) is a pure function of (seed, row, col), so any row range
can be generated independently on any GPU.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

SEED = 0x5A445701
VOCAB = 1 << 20

# the 200-column pattern: (sql type, count)
_MIX = [("varchar(255)", 100), ("int(11) unsigned", 40), ("bigint(20) unsigned", 20), ("smallint(5) unsigned", 20),
        ("tinyint(3) unsigned", 10), ("int(11)", 6), ("char(1)", 2), ("datetime", 1), ("decimal(24,12)", 1)]

# type ids as ReadDescFile assigns them (cplusplus/ConvertToZDW.cpp:117-157)
_TYPE_ID = {"varchar(255)": 0, "int(11) unsigned": 9, "bigint(20) unsigned": 10, "smallint(5) unsigned": 8,
            "tinyint(3) unsigned": 7, "int(11)": 14, "char(1)": 6, "datetime": 2, "decimal(24,12)": 11}
# generator class per sql type: 0 word, 1..4 integer of 7/15/31/47 bits, 5 char, 6 datetime, 7 decimal
_GEN = {"varchar(255)": 0, "tinyint(3) unsigned": 1, "smallint(5) unsigned": 2, "int(11) unsigned": 3, "int(11)": 3,
        "bigint(20) unsigned": 4, "char(1)": 5, "datetime": 6, "decimal(24,12)": 7}


def column_pattern():
    """200 sql types, the categories interleaved round-robin in proportion to their counts."""
    left = [n for _, n in _MIX]
    out = []
    acc = [0.0] * len(_MIX)
    for _ in range(200):
        # largest remaining share first (Bresenham-like interleave)
        for i, (_, n) in enumerate(_MIX):
            acc[i] += n / 200.0
        i = max(range(len(_MIX)), key=lambda k: (acc[k] if left[k] else -1e9))
        acc[i] -= 1.0
        left[i] -= 1
        out.append(_MIX[i][0])
    return out


ID_SQL = "varchar(64)"          # C5's high-cardinality id columns: 24-character practically unique tokens


def schema(ncols=200, id_every=0):
    """id_every = k > 0: every k-th column (k/2, k/2 + k, ...) is an id column -- C5 is schema(1000, id_every=20): 50 of them"""
    pat = column_pattern()
    sql = [ID_SQL if (id_every and i % id_every == id_every // 2) else pat[i % 200] for i in range(ncols)]
    desc = "".join("c%d\t%s\n" % (i, t) for i, t in enumerate(sql)).encode()
    return desc, [0 if t == ID_SQL else _TYPE_ID[t] for t in sql], [8 if t == ID_SQL else _GEN[t] for t in sql]


def _mix(x):
    x = np.asarray(x, dtype=np.uint64)
    x = (x ^ (x >> np.uint64(33))) * np.uint64(0xff51afd7ed558ccd)
    x = (x ^ (x >> np.uint64(33))) * np.uint64(0xc4ceb9fe1a85ec53)
    return x ^ (x >> np.uint64(33))


_ALPHA = np.frombuffer(b"abcdefghijklmnopqrstuvwxyz0123456789_", dtype=np.uint8)


def _word(i, seed):
    h = int(_mix(np.uint64(seed) ^ np.uint64(0x9E3779B97F4A7C15) * np.uint64(i + 1)))
    ln = 4 + h % 21
    cs = []
    for j in range(ln):
        h = int(_mix(np.uint64(h) + np.uint64(j + 1)))
        cs.append(_ALPHA[h % 37])
    return bytes(cs)


def make_table(seed=1, ncols=200, nrows=1000, vocab=4096, p_empty=0.25, p_repeat=0.3):
    """Returns (desc_text, type_ids, tsv_bytes).  numpy; fine for up to ~1e5 rows."""
    with np.errstate(over="ignore"):
        desc, types, gens = schema(ncols)
        rng = np.random.Generator(np.random.PCG64(seed))
        # Zipf(1.1) ranks over the vocabulary by inverse CDF on the exact weights
        w = 1.0 / np.arange(1, vocab + 1) ** 1.1
        cdf = np.cumsum(w / w.sum())
        words = [_word(i, seed) for i in range(vocab)]
        cols = []
        for c in range(ncols):
            g = gens[c]
            u = rng.random(nrows)
            empty = u < p_empty
            if g == 0:
                ranks = np.searchsorted(cdf, rng.random(nrows)).clip(0, vocab - 1)
                vals = [words[r] for r in ranks]
            elif g in (1, 2, 3, 4):
                bits = (7, 15, 31, 47)[g - 1]
                v = rng.integers(0, 1 << bits, nrows, dtype=np.uint64)
                vals = [b"%d" % x for x in v]
            elif g == 5:
                v = rng.integers(0, 26, nrows)
                vals = [bytes([97 + x]) for x in v]
            elif g == 6:
                v = rng.integers(0, 86400 * 28, nrows)
                vals = [b"2024-01-%02d %02d:%02d:%02d" % (1 + x // 86400, x // 3600 % 24, x // 60 % 60, x % 60) for x in v]
            else:
                a = rng.integers(0, 100000, nrows)
                b = rng.integers(0, 10 ** 12, nrows, dtype=np.uint64)
                vals = [b"%d.%012d" % (x, y) for x, y in zip(a, b)]
            rep = rng.random(nrows) < p_repeat
            col = []
            prev = b""
            for r in range(nrows):
                if r and rep[r]:
                    cur = prev
                else:
                    cur = b"" if empty[r] else vals[r]
                col.append(cur)
                prev = cur
            cols.append(col)
        rows = [b"\t".join(cols[c][r] for c in range(ncols)) + b"\n" for r in range(nrows)]
        return desc, types, b"".join(rows)


# ------------------------------------------------------------------------------------------------
# device generator
# ------------------------------------------------------------------------------------------------
LIB_SYNTH = os.path.join(_build.LIBDIR, "libzdw_synth.so")
_lib = None


def _synth_lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_SYNTH):
            raise RuntimeError("libzdw_synth.so is not built: " + LIB_SYNTH)
        L = C.CDLL(LIB_SYNTH)
        L.zdw_synth_row_lengths.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]
        L.zdw_synth_row_lengths.restype = C.c_int
        L.zdw_synth_fill.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.zdw_synth_fill.restype = C.c_int
        L.zdw_synth_set_empty.argtypes = [C.c_uint32, C.c_uint32]
        L.zdw_synth_set_empty.restype = C.c_int
        _lib = L
    return _lib


class DeviceTable:
    """Generates rows [row0, row0 + nrows) of the synthetic table into a torch uint8 CUDA tensor."""

    def __init__(self, ncols=200, seed=SEED, vocab=VOCAB, device="cuda", word_maxlen=24, p_empty=0.25, dead_cols=0.0, id_every=0):
        """word_maxlen: longest vocabulary word (24 = the benchmark's table; up to 279 for tables with long strings);
        p_empty: fraction of empty fields (0.25 = the benchmark's table); dead_cols: fraction of columns that are empty in
        every row (0 = the benchmark's table; the reference's analytics-hits: 0.88)"""
        import torch
        self.empty16 = int(round(p_empty * 65536))
        self.dead16 = int(round(dead_cols * 65536))
        self.torch = torch
        assert vocab < (1 << 24) + 1 and 24 <= word_maxlen <= 279
        vocab = min(vocab, (1 << 24) - 1) | ((word_maxlen - 24) << 24)
        self.ncols, self.seed, self.vocab, self.device = ncols, seed, vocab, device
        self.desc, self.types, self.gens = schema(ncols, id_every)
        self.d_gens = torch.tensor(self.gens, dtype=torch.uint8, device=device)

    def generate(self, row0, nrows):
        torch = self.torch
        L = _synth_lib()
        if L.zdw_synth_set_empty(self.empty16, self.dead16):
            raise RuntimeError("zdw_synth_set_empty failed")
        stream = torch.cuda.current_stream().cuda_stream
        lens = torch.empty(nrows + 1, dtype=torch.int64, device=self.device)
        rc = L.zdw_synth_row_lengths(self.d_gens.data_ptr(), self.ncols, self.seed, row0, nrows, self.vocab, lens.data_ptr(), stream)
        if rc:
            raise RuntimeError("zdw_synth_row_lengths: hip error %d" % rc)
        lens[nrows] = 0
        offs = torch.cumsum(lens, 0) - lens          # exclusive
        total = int(offs[nrows].item())
        out = torch.empty(total + 64, dtype=torch.uint8, device=self.device)
        rc = L.zdw_synth_fill(self.d_gens.data_ptr(), self.ncols, self.seed, row0, nrows, self.vocab, offs.data_ptr(), out.data_ptr(), stream)
        if rc:
            raise RuntimeError("zdw_synth_fill: hip error %d" % rc)
        torch.cuda.synchronize()
        return out, total

    def rows_for_bytes(self, target_bytes, probe_rows=4096):
        """How many rows give about target_bytes (from the mean row length of a probe)."""
        _, total = self.generate(0, probe_rows)
        return max(1, int(target_bytes * probe_rows / max(1, total)))
