"""CPU test (gloo, world_size 2 and 3) of the multi-block driver of a row-sharded input, zdw_amd/sharded.encode_blocks_dist: the
protocol -- strings and first positions to the owners, first-occurrence flags back, the arena walk handed from rank to rank, the
rank that holds the end cut back, later ranks carried into the next block, owners dropping what lies beyond the end -- run with
a pure-Python shard (tests/fake_shard.py) whose memory-limit model is scaled down to a few kilobytes.  The blocks must end
where a sequential walk over the whole input ends them, and the assembled file must be what the oracle writes when it is told
to end its blocks at those rows and fields."""
import os
import random
import struct

import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import fake_shard

NCOLS, ARENA, LIMIT = 5, 2048, 3


def _table(seed, nrows):
    rnd = random.Random(seed)
    vocab = [("w%d" % rnd.randrange(10 ** rnd.randrange(2, 9))).encode() for _ in range(40)]
    rows = []
    for i in range(nrows):
        f = []
        for c in range(NCOLS):
            x = rnd.random()
            f.append(b"" if x < 0.2 else (rnd.choice(vocab) if x < 0.6 else ("id%06d_%d_%s" % (i, c, "x" * rnd.randrange(0, 30))).encode()))
        rows.append(b"\t".join(f))
    return b"".join(r + b"\n" for r in rows), rows


def _sequential_splits(rows):
    """the reference's loop (ConvertToZDW.cpp:329-414) with the scaled-down model: [(rows in the block, fields of the next row visited)]"""
    out, seen, state, start, r = [], set(), (0, 0, 0), 0, 0
    while r < len(rows):
        tripped = None
        for c, f in enumerate(rows[r].split(b"\t")):
            if f and f not in seen:
                seen.add(f)
                state, hit = fake_shard.walk_arenas([f], ARENA, LIMIT, state)
                if hit is not None:
                    tripped = c + 1
                    break
        if tripped is not None:
            out.append((r - start, tripped))
            seen, state, start = set(), (0, 0, 0), r
            continue                         # the row starts the next block
        r += 1
    return out


def _worker(rank, world, port, parts, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from zdw_amd import sharded
    shard = fake_shard.FakeShard(parts[rank], NCOLS, ARENA)
    try:
        blocks = sharded.encode_blocks_dist(shard, rank, world, len(parts[rank]), LIMIT, device="cpu")
        q.put((rank, blocks))
    except Exception as e:  # noqa: BLE001
        q.put((rank, ("error", getattr(e, "code", None), getattr(e, "bad_row", None), str(e))))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, parts):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() * 7 + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, parts, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        r = q.get(timeout=180)
        got[r[0]] = r[1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def _cut(rows, fractions):
    cuts = [0] + [int(len(rows) * f) for f in fractions] + [len(rows)]
    return [b"".join(r + b"\n" for r in rows[a:b]) for a, b in zip(cuts, cuts[1:])]


@pytest.mark.parametrize("world,fractions", [(2, [0.5]), (2, [0.07]), (3, [0.3, 0.34]), (3, [0.0, 0.9])])
def test_blocks_of_a_row_sharded_input_end_where_the_sequential_walk_ends_them(world, fractions):
    from oracle import oracle
    tsv, rows = _table(5 + world, 700)
    splits = _sequential_splits(rows)
    assert len(splits) >= 3, "the table must make several blocks"
    got = _run(world, _cut(rows, fractions))
    assert all(isinstance(got[r], list) for r in range(world)), got
    nblocks = len(got[0])
    assert all(len(got[r]) == nblocks for r in range(world)) and nblocks == len(splits) + 1
    assert [b["num_rows"] for b in got[0][:-1]] == [s[0] for s in splits]
    assert [b["partial_fields"] for b in got[0][:-1]] == [s[1] for s in splits]
    desc = b"".join(b"c%d\tvarchar(255)\n" % c for c in range(NCOLS))
    rc, want, info = oracle.encode(desc, tsv, split_rows=[s[0] for s in splits], split_cols=[s[1] for s in splits])
    assert rc == 0 and info.nblocks == nblocks
    data = bytearray()
    for k in range(nblocks):
        per = [got[r][k] for r in range(world)]
        assert len(set((b["num_rows"], b["longest_line"], b["is_last"]) for b in per)) == 1
        st = next(b for b in per if b["col_size"])
        data += struct.pack("<IIB", per[0]["num_rows"], per[0]["longest_line"], 1 if per[0]["is_last"] else 0)
        data += b"".join(b["dict_piece"] for b in per)
        data += st["col_size"] + struct.pack("<%dQ" % len(st["col_base"]), *st["col_base"]) + b"".join(b["rows"] for b in per)
    assert bytes(data) == want[info.header_bytes:]


def test_a_bad_row_fails_the_block_it_falls_in():
    tsv, rows = _table(9, 500)
    splits = _sequential_splits(rows)
    first = splits[0][0]
    bad = first + 3                          # 0-based: the fourth row of the second block
    rows = list(rows)
    rows[bad] += b"\textra"
    got = _run(2, _cut(rows, [0.5]))
    for r in range(2):
        assert got[r][0] == "error" and got[r][1] == 15 and got[r][2] == bad - first + 1, got[r]
