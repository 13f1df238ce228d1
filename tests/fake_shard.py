"""A pure-Python stand-in for zdw_amd.sharded.EncoderPort (test infrastructure: the CPU / gloo test of the block driver's
protocol, tests/test_sharded_blocks_cpu.py).  It encodes tables of varchar columns only, without escapes or blank lines, and
speaks the driver's protocol -- partition / owner import / first flags / arena walk / trim / owner finish / offsets -- the way
the HIP library does, with a memory-limit model scaled down to bytes: a block ends when its strings take their `limit`-th arena
of `arena` bytes (stringheap.cpp:61-86 with a tiny arena), so a few kilobytes of text give several blocks."""
import bisect

import torch

IDENTITY_CAPS = [16384 << k for k in range(18)]
NO_MIN = (1 << 64) - 1


def key16(s):
    return (int.from_bytes(s[:8].ljust(8, b"\0"), "big"), int.from_bytes(s[8:16].ljust(8, b"\0"), "big"))


def bytes_for(v):
    n = 1
    while v >= 256:
        n += 1
        v //= 256
    return n


def walk_arenas(strings, arena, limit, state):
    """strings in the order they are first met; state = (arenas, entries, free).  Returns (state, index of the string whose arena
    is the limit-th, or None)."""
    arenas, entries, free = state
    for i, s in enumerate(strings):
        need = len(s) + 1
        entries += 1
        if need > free:
            arenas += 1
            free = max(need, arena)
            if arenas >= limit:
                return (arenas, entries, free - need), i
        free -= need
    return (arenas, entries, free), None


class FakeShard:
    def __init__(self, tsv, ncols, arena=4096):
        self.tsv, self.ncols, self.arena = tsv, ncols, arena
        self.lo = 0

    # ---- pass 1 ----
    def pass1(self):
        self.table, pos, bad = [], self.lo, 0
        for line in self.tsv[self.lo:].split(b"\n")[:-1]:
            fields, p = [], pos
            for f in line.split(b"\t"):
                fields.append((p, f))
                p += len(f) + 1
            self.table.append((pos, fields))
            if len(fields) != self.ncols and not bad:
                bad = len(self.table)
            pos += len(line) + 1
        self.kept, self.partial, self.next_lo = len(self.table), 0, len(self.tsv)
        self._index()
        return 0, len(self.table), bad, 1

    def _visited(self):
        for r, (_, fields) in enumerate(self.table[:self.kept + (1 if self.partial else 0)]):
            for c, (p, f) in enumerate(fields if r < self.kept else fields[:self.partial]):
                yield r, c, p, f

    def _index(self):
        self.first, self.seen = {}, [0] * self.ncols
        for r, c, p, f in self._visited():
            if f:
                if c < self.ncols:
                    self.seen[c] = 1
                if f not in self.first:
                    self.first[f] = (p, r, c)

    def model_counts(self):
        return len(self.first), sum(len(s) + 1 for s in self.first), max([len(s) for s in self.first] + [0])

    def export_ranges(self):
        return [NO_MIN] * self.ncols, [0] * self.ncols, list(self.seen)

    def import_ranges(self, mn, mx, seen):
        self.block_seen = list(seen)

    # ---- requester side of the dictionary ----
    def sample_keys(self, nsamples):
        srt = sorted(self.first)
        step = max(1, len(srt) // nsamples)
        return [key16(s) for s in srt[::step][:nsamples]]

    def partition(self, splitters, base, with_pos):
        nparts = len(splitters) + 1
        parts = [[] for _ in range(nparts)]
        for s, (p, _, _) in sorted(self.first.items(), key=lambda kv: kv[1][0]):
            parts[bisect.bisect_right(splitters, key16(s))].append((s, p))
        self.sent = [s for part in parts for s, _ in part]
        counts = [len(part) for part in parts]
        nb = [sum(len(s) + 1 for s, _ in part) for part in parts]
        lens = torch.tensor([len(s) for s in self.sent], dtype=torch.int32)
        byts = torch.tensor(list(b"".join(s + b"\0" for s in self.sent)), dtype=torch.uint8)
        pos = torch.tensor([base + p for part in parts for _, p in part], dtype=torch.int64) if with_pos else None
        return counts, nb, lens, byts, pos

    def model_walk(self, flags, limit, state):
        mine = sorted((s for s, f in zip(self.sent, flags.tolist()) if f), key=lambda s: self.first[s][0])
        state, i = walk_arenas(mine, self.arena, limit, state)
        if i is None:
            return state, None
        p, r, c = self.first[mine[i]]
        return state, (r, c + 1, self.table[r][0], p)

    def trim(self, row, fields, row_start):
        self.kept, self.partial, self.next_lo = row, fields, row_start
        self._index()
        return row

    # ---- owner side ----
    def owner_import(self, lens, byts, pos):
        raw, at, self.arrivals = bytes(byts.tolist()), 0, []
        for l in lens.tolist():
            self.arrivals.append(raw[at:at + l])
            at += l + 1
        self.arr_pos = pos.tolist() if pos is not None else None
        self.minpos = {}
        if self.arr_pos is not None:
            for s, p in zip(self.arrivals, self.arr_pos):
                self.minpos[s] = min(p, self.minpos.get(s, p))

    def owner_first_flags(self):
        return torch.tensor([1 if p == self.minpos[s] else 0 for s, p in zip(self.arrivals, self.arr_pos)], dtype=torch.int32)

    def owner_finish(self, last_pos):
        keep = sorted(set(s for s in self.arrivals if last_pos is None or self.minpos[s] <= last_pos))
        self.range_strings, self.local, at = keep, {}, 0
        for s in keep:
            self.local[s] = at
            at += len(s) + 1
        return at

    def owner_offsets(self, base):
        return torch.tensor([base + self.local.get(s, 0) for s in self.arrivals], dtype=torch.int32)

    def owner_section(self, dict_size, first_range):
        self.index_size = bytes_for(dict_size)
        body = b"".join(s + b"\0" for s in self.range_strings)
        if first_range:
            self.piece = b"\0" if dict_size <= 1 else bytes([self.index_size]) + dict_size.to_bytes(self.index_size, "little") + b"\0" + body
        else:
            self.piece = body

    def set_offsets(self, offs, dict_size, first_range):
        self.off = dict(zip(self.sent, offs.tolist()))
        self.owner_section(dict_size, first_range)

    # ---- rows ----
    def _stored(self, fields):
        return [self.off[f] if f else 0 for _, f in fields]

    def last_row(self):
        return self._stored(self.table[self.kept - 1][1]) if self.kept else [0] * self.ncols

    def line_caps(self):
        return list(IDENTITY_CAPS)

    def set_prev_row(self, prev):
        self.prev = list(prev) if prev is not None else [0] * self.ncols

    def dict_piece(self):
        return self.piece

    def stats(self):
        used = [c for c in range(self.ncols) if self.block_seen[c]]
        return bytes(self.index_size if self.block_seen[c] else 0 for c in range(self.ncols)), [0] * len(used)

    def rows_bytes(self):
        used = [c for c in range(self.ncols) if self.block_seen[c]]
        out, prev = bytearray(), self.prev
        for _, fields in self.table[:self.kept]:
            cur = self._stored(fields)
            flags, vals = bytearray((len(used) + 7) // 8), bytearray()
            for u, c in enumerate(used):
                if cur[c] != prev[c]:
                    flags[u // 8] |= 1 << (u % 8)
                    vals += cur[c].to_bytes(8, "little")[:self.index_size]
            out += flags + vals
            prev = cur
        return bytes(out)

    rows = rows_bytes          # (the port's name for it)

    def next_block(self):
        self.lo = self.next_lo
