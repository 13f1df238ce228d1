"""CPU tests (gloo, world_size 2) of the multi-GPU exchange in zdw_amd/sharded.py: the collectives
must give exactly what the pure merge functions give."""
import os
import random

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from zdw_amd import sharded


def _shard_data(rank, ncols=7):
    rnd = random.Random(100 + rank)
    mn = [rnd.choice([sharded.NO_MIN, rnd.getrandbits(64), rnd.randrange(1, 1000)]) for _ in range(ncols)]
    mx = [0 if m == sharded.NO_MIN else max(m, rnd.getrandbits(64)) for m in mn]
    seen = [rnd.randrange(2) for _ in range(ncols)]
    words = [bytes(rnd.choice(b"abcxyz\xff\\") for _ in range(rnd.randrange(1, 9))) for _ in range(5 + 3 * rank)]
    lens = torch.tensor([len(w) for w in words], dtype=torch.int32)
    byts = torch.tensor(list(b"".join(w + b"\0" for w in words)), dtype=torch.uint8)
    last = [rnd.getrandbits(64) for _ in range(ncols)]
    nrows = 0 if rank == 1 and ncols == 3 else 10 + rank
    return mn, mx, seen, lens, byts, last, nrows


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mn, mx, seen, lens, byts, last, nrows = _shard_data(rank)
    r = sharded.allreduce_ranges(mn, mx, seen, device="cpu")
    ol, ob = sharded.allgather_uniques(lens, byts, rank, world)
    prev = sharded.exchange_prev_row(last, nrows, rank, world, device="cpu")
    # all-to-all-v plumbing of the range-partitioned dictionary: rank r sends r + 1 + q elements to rank q
    send_counts = [rank + 1 + peer for peer in range(world)]
    payload = torch.tensor([1000 * rank + 10 * peer + k for peer in range(world) for k in range(send_counts[peer])], dtype=torch.int32)
    recv_counts = sharded.exchange_counts(send_counts, "cpu")
    two = sharded.exchange_counts(send_counts, "cpu", None, [3 * x + rank for x in send_counts])      # two lists, one all-to-all
    assert two[0] == recv_counts and two[1] == [3 * (peer + 1 + rank) + peer for peer in range(world)]
    got = sharded.alltoallv(payload, send_counts, recv_counts)
    gathered = sharded.allgather_i64([rank * 7, -rank], "cpu")
    # the status of a rank-local step travels with the range all-reduce; the line-buffer tables with the halo
    rs = sharded.allreduce_ranges(mn, mx, seen, device="cpu", status=15 if rank == 1 else 0)
    assert rs[:3] == r and rs[3] == 15
    caps = [max(16384 << k, 65536 if rank == 0 else 16384) for k in range(18)]       # rank 0 holds a 40 KB row
    prev2, longest = sharded.exchange_prev_row(last, nrows, rank, world, device="cpu", line_caps=caps)
    assert prev2 == prev and longest == 65536
    from zdw_amd import capi
    sharded.agree(0, "cpu", what="nothing")
    try:
        sharded._guarded(lambda: (_ for _ in ()).throw(capi.ZdwError(7, "rank-local")) if rank == 0 else 1, "cpu", None, "a step")
        raised = None
    except capi.ZdwError as e:
        raised = e.code
    assert raised == 7, raised          # both ranks: the one that failed and the one that did not
    # ... and without a collective of its own: the code of a rank-local step rides on the next data collective (the halo all-gather)
    out, code, err = sharded._local(lambda: (_ for _ in ()).throw(capi.ZdwError(9, "rank-local, late")) if rank == 1 else 5)
    assert (out, code) == ((5, 0) if rank == 0 else (None, 9))
    try:
        sharded.exchange_prev_row(last, nrows, rank, world, device="cpu", line_caps=caps, status=(code, err))
        late = None
    except capi.ZdwError as e:
        late = (e.code, "late" in str(e))
    assert late == ((9, True) if rank == 1 else (9, False)), late          # the failing rank raises its own error, the other the code it was sent
    prev3, longest3 = sharded.exchange_prev_row(last, nrows, rank, world, device="cpu", line_caps=caps, status=(0, None))
    assert prev3 == prev and longest3 == 65536
    q.put((rank, r, ol.tolist(), ob.tolist(), prev, recv_counts, got.tolist(), gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_collectives_match_pure_merges():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = {}
    for _ in range(world):
        rank, r, ol, ob, prev, rc_, a2a, gath = q.get(timeout=120)
        got[rank] = (r, ol, ob, prev, rc_, a2a, gath)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    data = [_shard_data(r) for r in range(world)]
    want_ranges = sharded.merge_ranges([(d[0], d[1], d[2]) for d in data])
    want_prev = sharded.chain_prev_rows([d[5] for d in data], [d[6] for d in data])
    for rank in range(world):
        r, ol, ob, prev, rc_, a2a, gath = got[rank]
        assert rc_ == [peer + 1 + rank for peer in range(world)]
        assert a2a == [1000 * peer + 10 * rank + k for peer in range(world) for k in range(peer + 1 + rank)]
        assert gath == [[p * 7, -p] for p in range(world)]
        assert (list(r[0]), list(r[1]), list(r[2])) == (want_ranges[0], want_ranges[1], want_ranges[2])
        wl, wb = sharded.concat_uniques([(d[3], d[4]) for d in data], skip=rank)
        assert ol == wl.tolist() and ob == wb.tolist()
        assert prev == want_prev[rank]


def test_pure_partition_helpers():
    assert sharded.pick_splitters([(5, 1), (3, 2), (9, 9), (1, 1), (7, 0), (2, 2)], 3) == [(3, 2), (7, 0)]
    assert sharded.pick_splitters([(1, 1)], 1) == []
    big = (1 << 64) - 1
    sp = sharded.pick_splitters([(big, big)] * 4 + [(2, 0), (2, 5)], 3)
    assert sp == sorted(sp) and len(sp) == 2
    assert sharded.range_bases([10, 0, 5]) == ([1, 11, 11], 16)
    assert sharded.range_bases([]) == ([], 1)
    for v in (0, 1, (1 << 63) - 1, 1 << 63, big):
        assert sharded._i64_to_u64(sharded._u64_to_i64(v)) == v


def test_pure_merges():
    mn, mx, seen = sharded.merge_ranges([([5, sharded.NO_MIN], [9, 0], [0, 0]), ([3, 7], [4, 1 << 63], [1, 0])])
    assert (mn, mx, seen) == ([3, 7], [9, 1 << 63], [1, 0])
    vals = [0, 1, (1 << 63) - 1, 1 << 63, (1 << 64) - 1]
    t = sharded.u64_to_ordered_i64(vals)
    assert t.tolist() == sorted(t.tolist())              # unsigned order survives as signed order
    assert sharded.ordered_i64_to_u64(t) == vals
    assert sharded.chain_prev_rows([[1], [2], [3]], [4, 0, 5]) == [None, [1], [1]]


def test_chain_longest_lines():
    """the reference's line buffer only grows: block i's header carries the largest of shards 0..i"""
    assert sharded.chain_longest_lines([16384, 65536, 16384, 32768, 131072, 16384]) == [16384, 65536, 65536, 65536, 131072, 131072]
    assert sharded.chain_longest_lines([]) == []


def test_compose_line_caps():
    """m_LongestLine of a row-sharded block: the shards' tables composed in row order (the buffer a shard starts with is
    what the shards before it left)."""
    same = [16384 << k for k in range(18)]                       # a shard without long rows leaves the buffer as it was
    grows = [max(16384 << k, 65536) for k in range(18)]          # a 40 KB row: at least 65536 afterwards
    assert sharded.compose_line_caps([same, same]) == [16384, 16384]
    assert sharded.compose_line_caps([same, grows, same]) == [16384, 65536, 65536]
    assert sharded.compose_line_caps([grows, [32768] + [16384 << k for k in range(1, 18)]]) == [65536, 65536]
    # the start matters: this shard doubles once more only when it starts from 16384 (a chunk boundary on an escaped newline)
    odd = [65536] + [16384 << k for k in range(1, 18)]
    assert sharded.compose_line_caps([odd]) == [65536] and sharded.compose_line_caps([[32768] + same[1:], odd]) == [32768, 32768]
    assert sharded.compose_line_caps([]) == []
