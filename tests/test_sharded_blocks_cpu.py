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


def _worker(rank, world, port, parts, q, fail=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    from zdw_amd import sharded
    shard = fake_shard.FakeShard(parts[rank], NCOLS, ARENA)
    if fail and rank == fail[1]:          # a rank-local step of one rank fails (from its `fail[2]`-th call on): every rank must raise
        from zdw_amd import capi
        real, calls = getattr(shard, fail[0]), [0]

        def boom(*a, **k):
            calls[0] += 1
            if calls[0] >= fail[2]:
                raise capi.ZdwError(20, "injected failure of " + fail[0])
            return real(*a, **k)
        setattr(shard, fail[0], boom)
    try:
        blocks = sharded.encode_blocks_dist(shard, rank, world, len(parts[rank]), LIMIT, device="cpu")
        q.put((rank, blocks))
    except Exception as e:  # noqa: BLE001
        q.put((rank, ("error", getattr(e, "code", None), getattr(e, "bad_row", None), str(e))))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, parts, fail=None):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30500 + (os.getpid() * 7 + world) % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, parts, q, fail)) for r in range(world)]
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


@pytest.mark.parametrize("step,rank,nth", [("rows", 1, 1), ("rows", 0, 1), ("stats", 1, 2), ("dict_piece", 0, 1), ("model_counts", 1, 1),
                                           ("export_ranges", 0, 2), ("import_ranges", 1, 1), ("set_prev_row", 1, 1), ("next_block", 0, 1),
                                           ("next_block", 1, 1)])
def test_a_rank_local_failure_in_a_block_stops_every_rank(step, rank, nth):
    """Every rank-local step of the block driver runs through _local() and its code rides on the next collective that exists anyway
    (the next block's first all-gather for pass 2 and what surrounds it): whichever step fails on one rank, in whichever block,
    no rank is left waiting in a collective (the process group's timeout is 60 s; the workers must be back long before) and every
    rank that still had a collective in front of it raises.  A failure behind the LAST collective of the last block concerns the
    failing rank alone."""
    tsv, rows = _table(7, 700)
    assert len(_sequential_splits(rows)) >= 3
    import time
    t0 = time.time()
    got = _run(2, _cut(rows, [0.5]), fail=(step, rank, nth))
    assert time.time() - t0 < 50, "a rank waited in a collective"
    assert got[rank][0] == "error" and got[rank][1] == 20 and "injected" in got[rank][3], got[rank]
    other = got[1 - rank]
    # (the other rank: the same code, unless the failure lay behind the input's last collective)
    assert (isinstance(other, tuple) and other[0] == "error" and other[1] == 20) or isinstance(other, list), other
    if rank == 0 and nth == 1:            # rank 0's first call of any step lies in the first of at least three blocks
        assert isinstance(other, tuple) and other[0] == "error" and other[1] == 20, other
