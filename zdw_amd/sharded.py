"""One ZDW block row-sharded over several GPUs.

Every shard (one zdw_ctx / capi.Encoder per GPU) holds a contiguous row range of the block.  The
block's bytes must not depend on the number of shards, so three things are exchanged after each
shard's pass 1:

  1. column ranges   min / max / seen per column     -> element-wise min / max / or   (all-reduce)
  2. dictionary      range-partitioned (default): 16-byte prefix samples of every shard's unique strings
                     are all-gathered and give N-1 splitters, strings travel to the owner of their key
                     range (all-to-all-v), the owner dedupes + sorts its range, range sizes are all-gathered and
                     every string's offset travels back (all-to-all-v): the node-wide dictionary is sorted once.
                     merge="allgather": every shard's unique strings -> union on every shard (all-gather-v);
                     each shard then sorts the same set, so offsets agree without a reverse map
  3. row halo        stored values of a shard's last row -> first row's "previous row" of the next shard

The exchange arithmetic lives in pure functions (merge_ranges, concat_uniques, chain_prev_rows)
shared by two drivers:
  * encode_shards_local : all shards in one process on one GPU (parity tests);
  * encode_shard_dist   : one shard per rank over torch.distributed -- RCCL on GPUs ("nccl"),
                          gloo in the CPU tests of the exchange itself.
"""
import torch

U64_SIGN = 1 << 63
NO_MIN = (1 << 64) - 1


# ------------------------------------------------------------------------------------------------
# pure exchange arithmetic
# ------------------------------------------------------------------------------------------------
def merge_ranges(parts):
    """parts: list of (min[], max[], seen[]) per shard -> element-wise (min, max, or).
    min uses 2^64-1 for "none" and max uses 0, exactly as pass 1 leaves them."""
    ncols = len(parts[0][0])
    mn = [min(p[0][c] for p in parts) for c in range(ncols)]
    mx = [max(p[1][c] for p in parts) for c in range(ncols)]
    seen = [1 if any(p[2][c] for p in parts) else 0 for c in range(ncols)]
    return mn, mx, seen


def u64_to_ordered_i64(vals):
    """unsigned order -> signed order (x xor 2^63), so that int64 MIN/MAX all-reduces compare as unsigned"""
    import numpy as np
    a = np.asarray(vals, dtype=np.uint64) ^ np.uint64(U64_SIGN)          # (numpy: this runs once per step and column on every rank)
    return torch.from_numpy(a.view(np.int64).copy())


def ordered_i64_to_u64(t):
    import numpy as np
    a = t.contiguous().numpy().view(np.uint64) ^ np.uint64(U64_SIGN)
    return a.tolist()


def concat_uniques(parts, skip=None):
    """parts: list of (lens int32/uint32 tensor[count], bytes uint8 tensor[nbytes]) per shard.
    Returns the concatenation over all shards except `skip` (a shard never imports its own strings)."""
    lens = [p[0] for i, p in enumerate(parts) if i != skip and p[0].numel()]
    byts = [p[1] for i, p in enumerate(parts) if i != skip and p[1].numel()]
    if not lens:
        dev = parts[0][0].device
        return torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.uint8, device=dev)
    return torch.cat(lens), torch.cat(byts)


def pick_splitters(samples, nparts):
    """samples: every shard's [(k0, k1)] 16-byte prefix keys (any order, all shards concatenated; a shard
    without strings contributes (2^64-1, 2^64-1)).  Returns nparts - 1 sorted splitters, the same on every
    shard because they are a pure function of the multiset of samples."""
    if nparts <= 1:
        return []
    inf = ((1 << 64) - 1, (1 << 64) - 1)
    srt = sorted(x for x in samples if tuple(x) != inf) or [inf]      # "no string here" samples carry no information
    n = len(srt)
    return [srt[min(n - 1, (j * n) // nparts)] for j in range(1, nparts)]


def range_bases(slice_bytes):
    """slice_bytes[r] = bytes of key range r.  Returns (dictionary offset of each range's first string, total
    dictionary size): offset 0 is the origin byte (dictionary.cpp:96)."""
    bases, at = [], 1
    for b in slice_bytes:
        bases.append(at)
        at += b
    return bases, at


def chain_prev_rows(last_rows, nrows):
    """last_rows[i]: stored values of shard i's last row (per column); nrows[i]: its row count.
    Returns prev[i] = the row that precedes shard i's first row (None for the block's first shard);
    shards without rows pass their predecessor's row through."""
    prev, cur = [], None
    for i, lr in enumerate(last_rows):
        prev.append(cur)
        if nrows[i]:
            cur = lr
    return prev


FIRST_ROW_CAP = 16384


def compose_line_caps(tables, start=FIRST_ROW_CAP):
    """tables[i][k] = the reference's line buffer after shard i's rows when it held 16384 << k before them
    (capi.Encoder.line_caps).  Returns the buffer after shards 0..i for every i: m_LongestLine of the block is the
    last entry (the buffer starts at 16384 -- `start`: what the blocks before left -- and only ever doubles,
    getnextrow.cpp:57-65)."""
    out, cap = [], start
    for t in tables:
        k = max(0, (int(cap) // FIRST_ROW_CAP).bit_length() - 1)
        cap = int(t[min(k, len(t) - 1)])
        out.append(cap)
    return out


# ------------------------------------------------------------------------------------------------
# device-pointer <-> torch plumbing
# ------------------------------------------------------------------------------------------------
class _RawCuda:
    def __init__(self, ptr, nbytes, typestr):
        itemsize = int(typestr[-1])
        self.__cuda_array_interface__ = {"shape": (nbytes // itemsize,), "typestr": typestr, "data": (ptr, False), "version": 2}


def _wrap(ptr, nbytes, typestr, device):
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8 if typestr == "|u1" else torch.int32, device=device)
    return torch.as_tensor(_RawCuda(ptr, nbytes, typestr), device=device)


def _torch_done():
    """The library works on its own HIP stream: anything torch produced for it must be complete first."""
    if torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def export_uniques_tensors(enc, device):
    """(lens int32[count], bytes uint8[nbytes]) as torch tensors owned by the caller (copies)."""
    d_lens, d_bytes, count, nbytes = enc.export_uniques()
    lens = _wrap(d_lens, count * 4, "<i4", device).clone()
    lens = lens[lens != 0]                 # the library lists uniques by id; ids it never used have length 0 and no bytes
    byts = _wrap(d_bytes, nbytes, "|u1", device).clone()
    _torch_done()          # the library may reuse its buffers as soon as we return
    return lens, byts


# ------------------------------------------------------------------------------------------------
# drivers
# ------------------------------------------------------------------------------------------------
def encode_shards_local(encoders, device="cuda"):
    """All shards in this process (shard order = row order).  Returns per-shard capi.Block list."""
    from . import capi
    p1s = []
    for e in encoders:
        rc, p1 = e.pass1()
        if rc:
            raise capi.ZdwError(rc, "pass 1 failed on a shard")
        p1s.append(p1)
    mn, mx, seen = merge_ranges([e.export_ranges() for e in encoders])
    uniq = [export_uniques_tensors(e, device) for e in encoders]
    for i, e in enumerate(encoders):
        e.import_ranges(mn, mx, seen)
        lens, byts = concat_uniques(uniq, skip=i)
        _torch_done()
        if lens.numel():
            e.import_uniques(lens.data_ptr(), byts.data_ptr(), lens.numel(), byts.numel())
    prev = chain_prev_rows([e.last_row() for e in encoders], [p.num_rows for p in p1s])
    longest = (compose_line_caps([e.line_caps() for e in encoders]) or [FIRST_ROW_CAP])[-1]      # one block: one header value
    blocks = []
    for i, e in enumerate(encoders):
        e.set_prev_row(prev[i])
        sec, _ = e.dict_section()
        cs, cb = e.stats()
        rows = e.pass2()
        blocks.append(capi.Block(p1s[i].num_rows, longest, True, sec, cs, cb, rows, p1s[i]))
    return blocks


# ------------------------------------------------------------------------------------------------
# Independent blocks: every shard's rows become their own ZDW block(s) with their own dictionary.
# No dictionary exchange at all; the file is NOT the one the reference writes for the whole input in one block
# (it is the one it writes when it ends a block at every shard boundary, which its memory limit may do anywhere),
# and every ZDW reader takes it: several blocks per file are part of the format.
def chain_longest_lines(own):
    """own[i] = m_LongestLine after shard i's rows alone.  The reference's line buffer only ever grows (doubling
    from 16384 until a row fits, ConvertToZDW.cpp:965), so after shard i it holds the largest of shards 0..i."""
    out, cur = [], 0
    for v in own:
        cur = max(cur, int(v))
        out.append(cur)
    return out


def encode_independent_blocks_one(enc, fetch=True):
    """All blocks of one shard's rows (normally one; more when its dictionary outgrows a block).
    Returns ([capi.Block] or [(num_rows, dict_len, rows_len)], [Pass1Result])."""
    from . import capi
    blocks, p1s = [], []
    while True:
        if fetch:
            rc, blk, p1 = enc.encode_block()
            if rc:
                raise capi.ZdwError(rc, "block encode failed")
            if blk is None:
                break
            blocks.append(blk)
        else:
            rc, p1 = enc.pass1()
            if rc:
                raise capi.ZdwError(rc, "pass 1 failed")
            if p1.num_rows == 0:
                break
            _, dlen, _ = enc.dict_section_device()
            enc.stats()
            _, rlen = enc.pass2_device()
            blocks.append((p1.num_rows, dlen, rlen))
        p1s.append(p1)
        if p1.is_last_block:
            break
        enc.next_block()
    return blocks, p1s


def encode_independent_blocks_local(encoders):
    """All shards in this process.  Returns the file's blocks in order (capi.Block, `is_last` and `longest_line`
    as the one file needs them)."""
    per = [encode_independent_blocks_one(e) for e in encoders]
    flat = [b for blocks, _ in per for b in blocks]
    caps = chain_longest_lines([b.longest_line for b in flat])
    for i, b in enumerate(flat):
        b.longest_line = caps[i]
        b.is_last = i + 1 == len(flat)
    return flat


def encode_independent_blocks_dist(enc, rank, world, group=None, fetch=False):
    """One rank's blocks.  One all-gather (block count and line-buffer size per rank) fixes the headers; the
    data path has no collective.  Returns the rank's blocks (capi.Block with final headers when fetch=True,
    else (num_rows, dict_len, rows_len) tuples with the outputs of the LAST block left in HBM)."""
    device = "cuda:%d" % torch.cuda.current_device()
    blocks, p1s = encode_independent_blocks_one(enc, fetch=fetch)
    own_cap = max([p.longest_line for p in p1s] + [0])
    allv = allgather_i64([len(blocks), own_cap], device, group=group)
    before = max([v[1] for v in allv[:rank]] + [0])
    last_rank_with_rows = max([r for r in range(world) if allv[r][0]] + [-1])
    if fetch:
        caps = chain_longest_lines([before] + [b.longest_line for b in blocks])[1:]
        for i, b in enumerate(blocks):
            b.longest_line = caps[i]
            b.is_last = rank == last_rank_with_rows and i + 1 == len(blocks)
    return blocks


SAMPLES_PER_SHARD = 256


def _split(t, counts):
    out, at = [], 0
    for c in counts:
        out.append(t[at:at + c])
        at += c
    return out


def encode_shards_partitioned_local(encoders, device="cuda"):
    """encode_shards_local with the range-partitioned dictionary: the all-to-all steps are plain tensor
    slicing here (all shards live in this process).  Returns per-shard capi.Block list; the blocks'
    dict_section fields are the PIECES of the section (concatenate them in shard order)."""
    from . import capi
    n = len(encoders)
    p1s = []
    for e in encoders:
        rc, p1 = e.pass1()
        if rc:
            raise capi.ZdwError(rc, "pass 1 failed on a shard")
        p1s.append(p1)
    mn, mx, seen = merge_ranges([e.export_ranges() for e in encoders])
    for e in encoders:
        e.import_ranges(mn, mx, seen)
    samples = []
    for e in encoders:
        e.sort_uniques()
        samples += e.sample_keys(SAMPLES_PER_SHARD)
    splitters = pick_splitters(samples, n)
    sent = []          # per shard: (counts, nbytes, lens tensor, bytes tensor)
    for e in encoders:
        counts, nb, d_lens, d_bytes = e.partition(splitters)
        lens = _wrap(d_lens, sum(counts) * 4, "<i4", device).clone()
        byts = _wrap(d_bytes, sum(nb), "|u1", device).clone()
        _torch_done()
        sent.append((counts, nb, _split(lens, counts), _split(byts, nb)))
    slices = []
    for r, e in enumerate(encoders):        # owner r receives part r of every shard
        lens = torch.cat([sent[q][2][r] for q in range(n)]) if n else None
        byts = torch.cat([sent[q][3][r] for q in range(n)])
        _torch_done()
        slices.append(e.owner_build(lens.data_ptr() if lens.numel() else 0, byts.data_ptr() if byts.numel() else 0, lens.numel(), byts.numel()))
    bases, total = range_bases(slices)
    replies = []
    for r, e in enumerate(encoders):
        cnt = sum(sent[q][0][r] for q in range(n))
        offs = _wrap(e.owner_offsets(bases[r]), cnt * 4, "<i4", device).clone()
        _torch_done()
        replies.append(_split(offs, [sent[q][0][r] for q in range(n)]))
    for q, e in enumerate(encoders):        # shard q gets back, range by range, the offsets of what it sent
        mine = torch.cat([replies[r][q] for r in range(n)])
        _torch_done()
        e.set_offsets(mine.data_ptr() if mine.numel() else 0, mine.numel(), total, q == 0)
    prev = chain_prev_rows([e.last_row() for e in encoders], [p.num_rows for p in p1s])
    longest = (compose_line_caps([e.line_caps() for e in encoders]) or [FIRST_ROW_CAP])[-1]      # one block: one header value
    blocks = []
    for i, e in enumerate(encoders):
        e.set_prev_row(prev[i])
        sec, _ = e.dict_section()
        cs, cb = e.stats()
        rows = e.pass2()
        blocks.append(capi.Block(p1s[i].num_rows, longest, True, sec, cs, cb, rows, p1s[i]))
    return blocks


def _staged(t, group):
    """gloo moves CPU tensors only for all-to-all; RCCL takes device tensors as they are"""
    import torch.distributed as dist
    return t.cpu() if dist.get_backend(group) == "gloo" and t.is_cuda else t


def alltoallv(t, send_counts, recv_counts, group=None):
    """1-D all-to-all-v: t = concatenation of the pieces for ranks 0..N-1 (send_counts elements each)."""
    import torch.distributed as dist
    dev = t.device
    src = _staged(t.contiguous(), group)
    out = torch.empty(int(sum(recv_counts)), dtype=t.dtype, device=src.device)
    dist.all_to_all_single(out, src, output_split_sizes=[int(x) for x in recv_counts], input_split_sizes=[int(x) for x in send_counts], group=group)
    return out.to(dev)


def exchange_counts(vals, device, group=None, *more):
    """vals[r] = a number for rank r.  Returns what every rank has for me: got[q] = rank q's vals[me].
    With further lists in `more` (same length) all of them travel in ONE all-to-all; returns a list per input."""
    import torch.distributed as dist
    lists = [vals] + list(more)
    k = len(lists)
    flat = [int(l[r]) for r in range(len(vals)) for l in lists]           # k numbers per destination rank
    t = torch.tensor(flat, dtype=torch.int64, device=device)
    src = _staged(t, group)
    out = torch.empty_like(src)
    dist.all_to_all_single(out, src, group=group)
    got = [int(x) for x in out.cpu().tolist()]
    res = [got[j::k] for j in range(k)]
    return res if more else res[0]


def allgather_i64(vals, device, group=None):
    """every rank's list of int64 (same length on all ranks), as a list of lists"""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    t = torch.tensor(vals, dtype=torch.int64, device=device)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t, group=group)
    return [[int(x) for x in o.cpu().tolist()] for o in outs]


def _u64_to_i64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _i64_to_u64(v):
    return v + (1 << 64) if v < 0 else v


class _Laps:
    """ZDW_DIST_LAPS=1: wall time of the phases of a distributed step on stderr (rank 0; each mark waits for the device)."""

    def __init__(self, rank):
        import os
        self.on = rank == 0 and os.environ.get("ZDW_DIST_LAPS") == "1"
        self.t, self.out = None, []
        self.mark(None)

    def mark(self, name):
        if not self.on:
            return
        import time
        torch.cuda.synchronize()
        now = time.perf_counter()
        if name is not None:
            self.out.append("%s %.2f" % (name, (now - self.t) * 1e3))
        self.t = now

    def done(self):
        if self.on:
            import sys
            print("[dist laps, ms] " + " | ".join(self.out), file=sys.stderr)


def _local(fn):
    """Runs a rank-local step; returns (its result or None, 0 or the error's code, the error).  The code travels with the NEXT collective
    (a rank that failed alone must not leave the others waiting in it): no collective of its own."""
    from . import capi
    try:
        return fn(), 0, None
    except capi.ZdwError as e:
        return None, (e.code or 20), e


def capi_error(code, text):
    from . import capi
    return capi.ZdwError(code, text)


def _raise_together(rcs, mine, err, what):
    """rcs: every rank's code for the same step (the same list on every rank).  Raises on all of them when one failed."""
    from . import capi
    if any(rcs):
        if err is not None:
            raise err
        raise capi.ZdwError(mine or max(rcs), "%s failed on a rank" % what)


def partitioned_dictionary(enc, rank, world, device, group=None):
    """The dictionary steps of one rank with the range-partitioned dictionary (after the range exchange).  Returns the return code of
    its last two rank-local steps (0, or the first failure's) and the error: the caller's next collective carries it (exchange_prev_row)."""
    laps = _Laps(rank)

    def _sample():
        enc.sort_uniques()
        return enc.sample_keys(SAMPLES_PER_SHARD)
    mine, rc, err = _local(_sample)
    flat = [_u64_to_i64(w) for k in (mine or [(0, 0)] * SAMPLES_PER_SHARD) for w in k] + [rc]
    allk = allgather_i64(flat, device, group=group)
    _raise_together([a[-1] for a in allk], rc, err, "sampling the dictionary")
    samples = [(_i64_to_u64(a[2 * i]), _i64_to_u64(a[2 * i + 1])) for a in allk for i in range((len(a) - 1) // 2)]
    splitters = pick_splitters(samples, world)
    laps.mark("sample+splitters")
    part, rc, err = _local(lambda: enc.partition(splitters))
    if part is not None:
        counts, nb, d_lens, d_bytes = part
        lens = _wrap(d_lens, sum(counts) * 4, "<i4", device).clone()      # torch-owned copies: the collectives work on memory torch allocated
        byts = _wrap(d_bytes, sum(nb), "|u1", device).clone()
        _torch_done()
    else:
        counts, nb = [0] * world, [0] * world
        lens, byts = torch.empty(0, dtype=torch.int32, device=device), torch.empty(0, dtype=torch.uint8, device=device)
    laps.mark("partition")
    rcounts, rbytes, rcs = exchange_counts(counts, device, group, nb, [rc] * world)
    _raise_together(rcs, rc, err, "partitioning the dictionary")
    got_lens = alltoallv(lens, counts, rcounts, group=group)
    got_byts = alltoallv(byts, nb, rbytes, group=group)
    _torch_done()
    laps.mark("strings all-to-all")
    # The owner's build in two halves.  First the arrivals are deduplicated: that fixes the bytes the range takes in the dictionary, so
    # the gather of those figures can come BEFORE the sort -- every rank then knows the dictionary's size, i.e. the width of its text
    # columns, and starts its row lengths on its second stream while the owners sort (the first row again once the halo row is known).
    # (an owner whose import fails still reports: the size all-gather carries its code, so nobody waits for it in vain)
    def _import():
        enc.owner_import(got_lens.data_ptr() if got_lens.numel() else 0, got_byts.data_ptr() if got_byts.numel() else 0, 0, got_lens.numel(), got_byts.numel())
        return enc.owner_size()
    slice_bytes, rc_own, err = _local(_import)
    gathered = allgather_i64([slice_bytes or 0, rc_own], device, group=group)
    _raise_together([a[1] for a in gathered], rc_own, err, "building a dictionary range")
    slices = [a[0] for a in gathered]
    laps.mark("owner import")
    bases, total = range_bases(slices)
    # (rank-local steps from here on: their codes go to the caller, whose next collective carries them)
    def _finish():
        if total < (1 << 32):
            enc.rows_early(total)
        got = enc.owner_finish(None)
        if got != slices[rank]:
            raise capi_error(20, "the range took %d bytes where %d were announced" % (got, slices[rank]))
        return enc.owner_offsets(bases[rank])
    d_offs, rc_tail, err_tail = _local(_finish)
    laps.mark("owner sort")
    if d_offs is not None:
        offs = _wrap(d_offs, sum(rcounts) * 4, "<i4", device).clone()
        _torch_done()
    else:
        offs = torch.zeros(int(sum(rcounts)), dtype=torch.int32, device=device)
    back = alltoallv(offs, rcounts, counts, group=group)
    _torch_done()
    laps.mark("offsets back")
    if rc_tail == 0:
        _, rc_tail, err_tail = _local(lambda: enc.set_offsets(back.data_ptr() if back.numel() else 0, back.numel(), total, rank == 0))
    laps.mark("set offsets")
    laps.done()
    return rc_tail, err_tail


def allreduce_ranges(mn, mx, seen, device, group=None, status=None):
    """RCCL/gloo form of merge_ranges: ONE all-reduce (max) over [~min, max, seen] -- ~x reverses the order of
    int64, so the maximum of the complements is the complement of the minimum.  With `status` (this rank's return
    code of pass 1) the same all-reduce carries it: returns (mn, mx, seen, worst status of any rank)."""
    import torch.distributed as dist
    tmn = u64_to_ordered_i64(mn)
    tmx = u64_to_ordered_i64(mx)
    n = tmn.numel()
    extra = [] if status is None else [int(status)]
    t = torch.cat([torch.bitwise_not(tmn), tmx, torch.tensor(list(seen) + extra, dtype=torch.int64)]).to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    t = t.cpu()
    res = (ordered_i64_to_u64(torch.bitwise_not(t[:n])), ordered_i64_to_u64(t[n:2 * n]), [int(x) for x in t[2 * n:3 * n].tolist()])
    return res if status is None else res + (int(t[3 * n].item()),)


def agree(rc, device, group=None, what="a shard step"):
    """Every rank learns the worst return code of a rank-local step and raises together: a rank that failed alone
    would leave the others waiting in the next collective until the RCCL timeout."""
    import torch.distributed as dist
    from . import capi
    t = torch.tensor([int(rc)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    worst = int(t.item())
    if worst:
        raise capi.ZdwError(worst, "%s failed on a rank (this rank: %d)" % (what, rc))


def _guarded(fn, device, group, what):
    """Run a rank-local step, then agree on its outcome (see agree)."""
    from . import capi
    rc, out, err = 0, None, None
    try:
        out = fn()
    except capi.ZdwError as e:
        rc, err = (e.code or 20), e
    try:
        agree(rc, device, group, what)
    except capi.ZdwError:
        if err is not None:
            raise err
        raise
    return out


def allgather_uniques(lens, byts, rank, world, group=None):
    """all-gather-v of (lens, bytes): sizes first, then padded payloads; returns the concatenation of the
    OTHER ranks' strings."""
    import torch.distributed as dist
    dev = lens.device
    sizes = torch.tensor([lens.numel(), byts.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    counts = [int(s[0].item()) for s in all_sizes]
    nbytes = [int(s[1].item()) for s in all_sizes]
    max_c, max_b = max(counts + [1]), max(nbytes + [1])
    pl = torch.zeros(max_c, dtype=torch.int32, device=dev)
    pl[:lens.numel()] = lens
    pb = torch.zeros(max_b, dtype=torch.uint8, device=dev)
    pb[:byts.numel()] = byts
    gl = [torch.empty(max_c, dtype=torch.int32, device=dev) for _ in range(world)]
    gb = [torch.empty(max_b, dtype=torch.uint8, device=dev) for _ in range(world)]
    dist.all_gather(gl, pl, group=group)
    dist.all_gather(gb, pb, group=group)
    parts = [(gl[r][:counts[r]], gb[r][:nbytes[r]]) for r in range(world)]
    return concat_uniques(parts, skip=rank)


def exchange_prev_row(last_row, nrows, rank, world, device, group=None, line_caps=None, start_cap=FIRST_ROW_CAP, first=0, status=None):
    """Row halo: every rank learns the stored values that precede its first row.  With `line_caps` (this rank's
    capi.Encoder.line_caps()) the same all-gather carries the tables and the block's m_LongestLine is returned too.
    status = (code, error) of this rank's steps since the last collective: the all-gather carries the code, and every rank raises
    when one of them failed."""
    import torch.distributed as dist
    ncols = len(last_row)
    caps = list(line_caps) if line_caps is not None else []
    mine = torch.empty(ncols + 1 + len(caps) + (1 if status is not None else 0), dtype=torch.int64)
    mine[0] = nrows
    if ncols:
        mine[1:ncols + 1] = u64_to_ordered_i64(last_row)
    if caps:
        mine[ncols + 1:ncols + 1 + len(caps)] = torch.tensor(caps, dtype=torch.int64)
    if status is not None:
        mine[-1] = int(status[0])
    mine = mine.to(device)
    allr = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine, group=group)
    allr = [t.cpu() for t in allr]
    if status is not None:
        _raise_together([int(t[-1].item()) for t in allr], status[0], status[1], "the dictionary")
        allr = [t[:-1] for t in allr]
    rows = [ordered_i64_to_u64(t[1:ncols + 1]) for t in allr]
    counts = [int(t[0].item()) for t in allr]
    # (`first`: the ranks before it hold none of the block's rows and say nothing about the row before its first one)
    prev = ([None] * first + chain_prev_rows(rows[first:], counts[first:]))[rank]
    if line_caps is None:
        return prev
    return prev, compose_line_caps([[int(x) for x in t[ncols + 1:].tolist()] for t in allr], start=start_cap)[-1]


def encode_shard_dist(enc, rank, world, group=None, fetch=False, merge=None):
    """One rank's part of the block.  Returns (num_rows, dict_section_len, rows_len) with the outputs
    left in HBM, or host copies when fetch=True.  merge: "partition" (range-partitioned dictionary: this
    rank's dict section is the piece it owns) or "allgather" (every rank holds the whole dictionary);
    default from ZDW_SHARD_MERGE, else "partition"."""
    import os
    if merge is None:
        merge = os.environ.get("ZDW_SHARD_MERGE", "partition")
    from . import capi
    device = "cuda:%d" % torch.cuda.current_device()
    rc, p1 = enc.pass1()
    # A rank-local failure (a row with a wrong field count, a NUL byte, memory) must fail the job on EVERY rank before
    # the first data collective, or the others wait in it until the RCCL timeout: the range all-reduce carries the code.
    ncols = enc.ncols
    mine = enc.export_ranges() if rc == 0 else ([NO_MIN] * ncols, [0] * ncols, [0] * ncols)
    mn, mx, seen, worst = allreduce_ranges(*mine, device=device, group=group, status=rc)
    if worst:
        raise capi.ZdwError(rc or worst, "pass 1 failed on rank %d" % rank if rc else "pass 1 failed on another rank")
    enc.import_ranges(mn, mx, seen)
    rc_tail, err_tail = 0, None
    if merge == "partition":
        rc_tail, err_tail = partitioned_dictionary(enc, rank, world, device, group=group)
    else:
        lens, byts = export_uniques_tensors(enc, device)
        olens, obyts = allgather_uniques(lens, byts, rank, world, group=group)
        _torch_done()
        _, rc_tail, err_tail = _local(lambda: enc.import_uniques(olens.data_ptr(), obyts.data_ptr(), olens.numel(), obyts.numel()) if olens.numel() else None)
    # (the row halo's all-gather carries the code of everything rank-local since the last collective)
    halo, rc_h, err_h = _local(lambda: (enc.last_row(), enc.line_caps())) if rc_tail == 0 else (None, rc_tail, err_tail)
    last, caps = halo if halo is not None else ([0] * ncols, list(IDENTITY_CAPS))
    prev, longest = exchange_prev_row(last, p1.num_rows, rank, world, device, group=group, line_caps=caps, status=(rc_h, err_h))
    enc.set_prev_row(prev)
    if fetch:
        sec, _ = enc.dict_section()
        cs, cb = enc.stats()
        rows = enc.pass2()
        return capi.Block(p1.num_rows, longest, True, sec, cs, cb, rows, p1)
    _, dlen, _ = enc.dict_section_device()
    enc.stats()
    _, rlen = enc.pass2_device()
    return p1.num_rows, dlen, rlen


# ------------------------------------------------------------------------------------------------
# Several blocks of a row-sharded input: where the reference's memory limit ends a block (SURVEY.md 8e "Invariance
# requirement", Q9; stringheap.cpp:74-86, memory.cpp:76-81, ConvertToZDW.cpp:338,404,765-894).  The decision is taken on the
# node-wide dictionary: the dictionary exchange pauses once the owners hold every string with the position of its first
# occurrence; they tell the requesters which of their occurrences are the first node-wide; the ranks walk the reference's arenas
# over their own first occurrences in row order (the state goes from rank to rank); the rank in which the limit trips cuts its
# part of the block back to that field, the ranks behind it keep their rows for the next block, and the owners drop what was
# first met beyond the end before they sort.  zdw_multi_encode (csrc/zdw_multi.hip) is the same protocol inside one process.
# ------------------------------------------------------------------------------------------------
IDENTITY_CAPS = [FIRST_ROW_CAP << k for k in range(18)]          # the line-buffer table of a rank without rows in the block


class EncoderPort:
    """What the block driver needs from a shard, in torch tensors (the collectives move those): a thin adapter over
    capi.Encoder (created with track_first_occurrence and defer_bad_row).  tests/ has a pure-Python stand-in with the same
    methods for the CPU (gloo) test of the protocol."""

    def __init__(self, enc, device):
        self.enc, self.device, self.ncols = enc, device, enc.ncols
        self._sent = 0

    def pass1(self):
        rc, p1 = self.enc.pass1()
        return rc, p1.num_rows, p1.bad_row, p1.is_last_block

    def model_counts(self):
        return self.enc.model_counts()

    def export_ranges(self):
        return self.enc.export_ranges()

    def import_ranges(self, mn, mx, seen):
        self.enc.import_ranges(mn, mx, seen)

    def sample_keys(self, nsamples):
        self.enc.sort_uniques()
        return self.enc.sample_keys(nsamples)

    def partition(self, splitters, base, with_pos):
        counts, nb, d_lens, d_bytes = self.enc.partition(splitters)
        self._sent = sum(counts)
        lens = _wrap(d_lens, sum(counts) * 4, "<i4", self.device).clone()
        byts = _wrap(d_bytes, sum(nb), "|u1", self.device).clone()
        pos = None
        if with_pos:
            pos = _wrap(self.enc.positions(base), sum(counts) * 8, "<i8", self.device).clone() if sum(counts) else torch.empty(0, dtype=torch.int64, device=self.device)
        _torch_done()
        return counts, nb, lens, byts, pos

    def owner_import(self, lens, byts, pos):
        _torch_done()
        if pos is not None and not pos.numel():
            # positions are expected and none arrived: the library still wants a pointer it may read -- eight real bytes of device
            # memory, not a sentinel (a wild address if anything ever looks at it)
            pos = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._keep = (lens, byts, pos)
        self._received = lens.numel()
        self.enc.owner_import(lens.data_ptr() if lens.numel() else 0, byts.data_ptr() if byts.numel() else 0,
                              pos.data_ptr() if pos is not None else 0, lens.numel(), byts.numel())

    def owner_first_flags(self):
        t = _wrap(self.enc.owner_first_flags(), self._received * 4, "<i4", self.device).clone()
        _torch_done()
        return t

    def model_walk(self, flags, limit_mb, state):
        _torch_done()
        return self.enc.model_walk(flags.data_ptr() if flags.numel() else 0, flags.numel(), limit_mb, state)

    def trim(self, row, fields, row_start):
        return self.enc.trim(row, fields, row_start).num_rows

    def owner_finish(self, last_pos):
        return self.enc.owner_finish(last_pos)

    def owner_offsets(self, base):
        t = _wrap(self.enc.owner_offsets(base), self._received * 4, "<i4", self.device).clone()
        _torch_done()
        return t

    def set_offsets(self, offs, dict_size, first_range):
        _torch_done()
        self.enc.set_offsets(offs.data_ptr() if offs.numel() else 0, offs.numel(), dict_size, first_range)

    def owner_section(self, dict_size, first_range):
        self.enc.owner_section(dict_size, first_range)

    def last_row(self):
        return self.enc.last_row()

    def line_caps(self):
        return self.enc.line_caps()

    def set_prev_row(self, prev):
        self.enc.set_prev_row(prev)

    def dict_piece(self):
        return self.enc.dict_section()[0]

    def stats(self):
        return self.enc.stats()

    def rows(self):
        return self.enc.pass2()

    def next_block(self):
        self.enc.next_block()


def broadcast_i64(vals, src, device, group=None):
    """rank `src`'s list of int64 on every rank (the others pass a list of the same length)"""
    import torch.distributed as dist
    t = torch.tensor([int(v) for v in vals], dtype=torch.int64, device=device)
    dist.broadcast(t, src=src, group=group)
    return [int(x) for x in t.cpu().tolist()]


def encode_blocks_dist(port, rank, world, nbytes, mem_limit_mb, group=None, device="cpu"):
    """All blocks of an input whose rows lie on the ranks in rank order (rank r holds `nbytes` bytes of it: whole rows).
    Returns this rank's share of every block: [dict(num_rows=block's rows, longest_line, is_last, dict_piece, col_size,
    col_base, rows)] -- the block's dictionary section is the ranks' pieces in rank order, its rows are the ranks' rows in rank
    order (a rank without rows in a block contributes b"")."""
    from . import capi
    sizes = [a[0] for a in allgather_i64([nbytes], device, group=group)]
    bases = [sum(sizes[:r]) for r in range(world)]
    first, have_p1, longest = 0, False, FIRST_ROW_CAP
    my_rows = my_bad = 0
    out = []
    # Rank-local steps (anything that can raise on one rank alone: memory in pass 2, a device error) run through _local(); their
    # code travels with the NEXT collective that exists anyway, so a rank that failed alone never leaves the others waiting in it.
    # carry: the code (and error) of the steps behind the last collective of the block before -- it rides on this block's first one.
    carry_rc, carry_err = 0, None
    while True:
        # ---- pass 1 where this rank still holds rows and has none ----
        rc, err = carry_rc, carry_err
        if rc == 0 and rank >= first and not have_p1:
            res, rc, err = _local(port.pass1)
            if rc == 0:
                rc, my_rows, my_bad, is_last = res
                if rc == 0 and not is_last:
                    rc = capi.OUT_OF_MEMORY                   # the rank's rows do not fit its device as one block
            have_p1 = rc == 0
        cnts = (0, 0, 0)
        if rank >= first and rc == 0 and mem_limit_mb:
            res, rc, err = _local(port.model_counts)
            if rc == 0:
                cnts = res
        info = allgather_i64([rc, my_rows if rank >= first else 0, my_bad if rank >= first else 0] + list(cnts), device, group=group)
        if any(a[0] for a in info):
            if err is not None:
                raise err
            raise capi.ZdwError(rc or max(a[0] for a in info), "pass 1 (or the rows of the block before) failed on a rank")
        rows_of = [a[1] for a in info]
        bad_rank = next((r for r in range(first, world) if info[r][2]), None)
        # Not every rank that still holds rows has to send its strings: the block ends where the dictionary has grown to the limit, so
        # the ranks are taken in row order until their strings -- counted as if none were shared, an upper bound -- reach twice the
        # limit; if the limit then does not trip among them, twice as many take part in a second attempt (csrc/zdw_multi.hip does
        # the same).  Every rank computes the same window from the gathered counts.
        window = 1 if mem_limit_mb else world
        while True:
            while True:
                last = min(world - 1, first + window - 1)
                ub = capi.model_upper_bound_mb(sum(a[3] for a in info[first:last + 1]), sum(a[4] for a in info[first:last + 1]),
                                               max(a[5] for a in info[first:last + 1])) if mem_limit_mb else 0.0
                if not mem_limit_mb or last == world - 1 or ub >= 2.0 * mem_limit_mb:
                    break
                window *= 2
            maybe = bool(mem_limit_mb) and ub >= mem_limit_mb
            sends = first <= rank <= last
            bad_rank = next((r for r in range(first, last + 1) if info[r][2]), None)

            # ---- the strings travel to the owners of their key ranges ----
            mine = _guarded(lambda: port.sample_keys(SAMPLES_PER_SHARD) if sends else [], device, group, "sampling the dictionary")
            flat = [_u64_to_i64(w) for k in mine for w in k] + [0] * (2 * SAMPLES_PER_SHARD - 2 * len(mine))
            allk = allgather_i64([len(mine)] + flat, device, group=group)
            samples = [(_i64_to_u64(a[1 + 2 * i]), _i64_to_u64(a[2 + 2 * i])) for a in allk for i in range(a[0])]
            splitters = pick_splitters(samples, world)
            if sends:
                counts, nb, lens, byts, pos = _guarded(lambda: port.partition(splitters, bases[rank], maybe), device, group, "partitioning the dictionary")
            else:
                _guarded(lambda: None, device, group, "partitioning the dictionary")
                counts, nb = [0] * world, [0] * world
                lens, byts = torch.empty(0, dtype=torch.int32, device=device), torch.empty(0, dtype=torch.uint8, device=device)
                pos = torch.empty(0, dtype=torch.int64, device=device) if maybe else None
            rcounts, rbytes = exchange_counts(counts, device, group, nb)
            got_lens = alltoallv(lens, counts, rcounts, group=group)
            got_byts = alltoallv(byts, nb, rbytes, group=group)
            got_pos = alltoallv(pos, counts, rcounts, group=group) if maybe else None
            _guarded(lambda: port.owner_import(got_lens, got_byts, got_pos), device, group, "building a dictionary range")

            # ---- where does the block end? ----
            end, trip = last, None
            if maybe:
                flags = _guarded(port.owner_first_flags, device, group, "first occurrences")
                my_flags = alltoallv(flags, rcounts, counts, group=group)
                state = (0, 0, 0)
                for i in range(first, last + 1):                   # in row order: the state goes from rank to rank
                    msg = [0] * 9
                    if rank == i:
                        try:
                            state, hit = port.model_walk(my_flags, mem_limit_mb, state)
                            msg = [0] + list(state) + ([1] + list(hit) if hit else [0, 0, 0, 0, 0])
                        except capi.ZdwError as e:
                            msg[0] = e.code or 20
                    msg = broadcast_i64(msg, i, device, group=group)
                    if msg[0]:
                        raise capi.ZdwError(msg[0], "the block-limit walk failed on rank %d" % i)
                    state = tuple(msg[1:4])
                    if msg[4]:
                        trip = (msg[5], msg[6], msg[7], bases[i] + msg[8])      # row, fields, row_start (in rank i's image), position in the input
                        end = i
                        break
            if trip is not None or last == world - 1:
                break                                          # the block ends among these ranks, or takes all the rows that are left
            window *= 2                                        # (else: the limit does not trip among them: twice as many ranks)
        if bad_rank is not None:
            # the reference meets the rows in order: it reports the bad row unless the block ends before it (ConvertToZDW.cpp:336-338)
            bad0 = info[bad_rank][2] - 1
            if trip is None or bad_rank < end or (bad_rank == end and bad0 <= trip[0]):
                e = capi.ZdwError(capi.WRONG_NUM_OF_COLUMNS_ON_A_ROW, "Row %u had the problem" % (sum(rows_of[first:bad_rank]) + info[bad_rank][2]))
                e.bad_row = sum(rows_of[first:bad_rank]) + info[bad_rank][2]
                raise e
        if trip is not None:
            kept = _guarded(lambda: port.trim(trip[0], trip[1], trip[2]) if rank == end else 0, device, group, "cutting the block back")
            rows_of[end] = broadcast_i64([kept], end, device, group=group)[0]
            if rank == end:
                my_rows = rows_of[end]
        in_block = first <= rank <= end
        block_rows = sum(rows_of[first:end + 1])
        if trip is not None and block_rows == 0:
            raise capi.ZdwError(capi.OUT_OF_MEMORY, "block limit leaves no room for a single row")

        # ---- column ranges of the block's ranks ----
        ncols = port.ncols
        no_ranges = ([NO_MIN] * ncols, [0] * ncols, [0] * ncols)
        mine_r, rc_r, err_r = _local(port.export_ranges) if in_block else (no_ranges, 0, None)
        mn, mx, seen, worst = allreduce_ranges(*(mine_r if rc_r == 0 else no_ranges), device=device, group=group, status=rc_r)
        if worst:
            if err_r is not None:
                raise err_r
            raise capi.ZdwError(worst, "the column ranges failed on a rank")

        # ---- the owners sort and lay out what belongs to the block; every string's offset travels back ----
        # (the gather of the range sizes also carries the code of installing the ranges)
        _, rc_own, err_own = _local(lambda: port.import_ranges(mn, mx, seen)) if in_block else (None, 0, None)
        slice_bytes = 0
        if rc_own == 0:
            slice_bytes, rc_own, err_own = _local(lambda: port.owner_finish(trip[3] if trip is not None else None))
            slice_bytes = slice_bytes or 0
        gathered = allgather_i64([slice_bytes, rc_own], device, group=group)
        if any(a[1] for a in gathered):
            if err_own is not None:
                raise err_own
            raise capi.ZdwError(rc_own or max(a[1] for a in gathered), "building a dictionary range failed on a rank")
        range_base, total = range_bases([a[0] for a in gathered])
        if total >= 1 << 32:
            raise capi.ZdwError(capi.OUT_OF_MEMORY, "the block's dictionary exceeds 2^32 bytes")
        offs = _guarded(lambda: port.owner_offsets(range_base[rank]), device, group, "dictionary offsets")
        back = alltoallv(offs, rcounts, counts, group=group)
        _guarded(lambda: port.set_offsets(back, total, rank == 0) if in_block else port.owner_section(total, rank == 0), device, group, "dictionary offsets")

        # ---- row halo and line buffer over the block's ranks ----
        last, caps = _guarded(lambda: (port.last_row(), port.line_caps()) if in_block else ([0] * ncols, IDENTITY_CAPS), device, group, "the row halo")
        prev, longest = exchange_prev_row(last, my_rows if in_block else 0, rank, world, device, group=group, line_caps=caps, start_cap=longest,
                                          first=first)
        # ---- this rank's share of the block: nothing below takes part in a collective, and the next one is the next block's first
        # ---- (a rank that fails here must reach it and say so: pass 2 is where a device runs out of memory)
        def finish():
            if in_block:
                port.set_prev_row(prev)
            piece = port.dict_piece()
            cs, cb = port.stats() if in_block else (b"", [])
            rows = port.rows() if in_block else b""
            if trip is not None and rank == end:
                port.next_block()
            return piece, cs, cb, rows
        res, carry_rc, carry_err = _local(finish)
        if trip is None:
            if carry_err is not None:
                raise carry_err                                # the last block: no collective follows, nobody waits for this rank
            piece, cs, cb, rows = res
        else:
            piece, cs, cb, rows = res if carry_rc == 0 else (b"", b"", [], b"")
        out.append(dict(num_rows=block_rows, longest_line=longest, is_last=trip is None, dict_piece=piece, col_size=cs, col_base=cb, rows=rows,
                        my_rows=my_rows if in_block else 0, partial_fields=trip[1] if trip is not None else 0))
        if trip is None:
            return out
        if rank == end:
            have_p1 = False
        first = end
