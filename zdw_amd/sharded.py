"""One ZDW block row-sharded over several GPUs.

Every shard (one zdw_ctx / capi.Encoder per GPU) holds a contiguous row range of the block.  The
block's bytes must not depend on the number of shards, so three things are exchanged after each
shard's pass 1:

  1. column ranges   min / max / seen per column     -> element-wise min / max / or   (all-reduce)
  2. dictionary      every shard's unique strings    -> union on every shard          (all-gather-v)
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
    t = torch.tensor([(v ^ U64_SIGN) - (1 << 64) if (v ^ U64_SIGN) >= U64_SIGN else (v ^ U64_SIGN) for v in vals], dtype=torch.int64)
    return t


def ordered_i64_to_u64(t):
    return [((int(x) & ((1 << 64) - 1)) ^ U64_SIGN) for x in t.tolist()]


def concat_uniques(parts, skip=None):
    """parts: list of (lens int32/uint32 tensor[count], bytes uint8 tensor[nbytes]) per shard.
    Returns the concatenation over all shards except `skip` (a shard never imports its own strings)."""
    lens = [p[0] for i, p in enumerate(parts) if i != skip and p[0].numel()]
    byts = [p[1] for i, p in enumerate(parts) if i != skip and p[1].numel()]
    if not lens:
        dev = parts[0][0].device
        return torch.empty(0, dtype=torch.int32, device=dev), torch.empty(0, dtype=torch.uint8, device=dev)
    return torch.cat(lens), torch.cat(byts)


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
    blocks = []
    for i, e in enumerate(encoders):
        e.set_prev_row(prev[i])
        sec, _ = e.dict_section()
        cs, cb = e.stats()
        rows = e.pass2()
        blocks.append(capi.Block(p1s[i].num_rows, p1s[i].longest_line, True, sec, cs, cb, rows, p1s[i]))
    return blocks


def allreduce_ranges(mn, mx, seen, device, group=None):
    """RCCL/gloo form of merge_ranges."""
    import torch.distributed as dist
    tmn = u64_to_ordered_i64(mn).to(device)
    tmx = u64_to_ordered_i64(mx).to(device)
    ts = torch.tensor(seen, dtype=torch.int32, device=device)
    dist.all_reduce(tmn, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(tmx, op=dist.ReduceOp.MAX, group=group)
    dist.all_reduce(ts, op=dist.ReduceOp.MAX, group=group)
    return ordered_i64_to_u64(tmn.cpu()), ordered_i64_to_u64(tmx.cpu()), [int(x) for x in ts.cpu().tolist()]


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


def exchange_prev_row(last_row, nrows, rank, world, device, group=None):
    """Row halo: every rank learns the stored values that precede its first row."""
    import torch.distributed as dist
    ncols = len(last_row)
    mine = torch.empty(ncols + 1, dtype=torch.int64, device=device)
    mine[0] = nrows
    if ncols:
        mine[1:] = u64_to_ordered_i64(last_row).to(device)
    allr = [torch.empty(ncols + 1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(allr, mine, group=group)
    rows = [ordered_i64_to_u64(t[1:].cpu()) for t in allr]
    counts = [int(t[0].item()) for t in allr]
    return chain_prev_rows(rows, counts)[rank]


def encode_shard_dist(enc, rank, world, group=None, fetch=False):
    """One rank's part of the block.  Returns (num_rows, dict_section_len, rows_len) with the outputs
    left in HBM, or host copies when fetch=True."""
    from . import capi
    device = "cuda:%d" % torch.cuda.current_device()
    rc, p1 = enc.pass1()
    if rc:
        raise capi.ZdwError(rc, "pass 1 failed on rank %d" % rank)
    mn, mx, seen = allreduce_ranges(*enc.export_ranges(), device=device, group=group)
    enc.import_ranges(mn, mx, seen)
    lens, byts = export_uniques_tensors(enc, device)
    olens, obyts = allgather_uniques(lens, byts, rank, world, group=group)
    _torch_done()
    if olens.numel():
        enc.import_uniques(olens.data_ptr(), obyts.data_ptr(), olens.numel(), obyts.numel())
    prev = exchange_prev_row(enc.last_row(), p1.num_rows, rank, world, device, group=group)
    enc.set_prev_row(prev)
    if fetch:
        sec, _ = enc.dict_section()
        cs, cb = enc.stats()
        rows = enc.pass2()
        return capi.Block(p1.num_rows, p1.longest_line, True, sec, cs, cb, rows, p1)
    _, dlen, _ = enc.dict_section_device()
    enc.stats()
    _, rlen = enc.pass2_device()
    return p1.num_rows, dlen, rlen
