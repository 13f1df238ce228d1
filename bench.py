#!/usr/bin/env python3
"""bench.py — TSV -> ZDW encode throughput on MI355X (BASELINE.json metric).

One step = one full two-pass block encode (row scan, pass 1, dictionary sort, column stats, pass 2)
of a synthetic TSV that is already resident in HBM; outputs stay in HBM.  Workload at N = 1:
BASELINE.json configs[3], "Synthetic 10 GB TSV, 200 cols mixed VARCHAR/LONG, Zipf strings"
(zdw_amd/synth.py).  N > 1 (launched by torch.distributed.run, one rank per GPU):
weak scaling, every rank holds its own 10 GB row range of ONE ZDW block; RCCL carries the column
range all-reduce, the dictionary merge (range-partitioned: two all-to-all-v of unique strings / offsets,
small all-gathers of samples and range sizes) and the previous-row halo.

After the timed region (never inside it) rank 0 also: verifies the last step's outputs ("verified": the bytes the compiled
reference wrote for the CPU baseline's sample -- the head of the same table -- equal to the HIP encode of those rows; row count,
dictionary header + strict strcmp order of every entry, a walk of the first rows, and byte equality of the
dictionary / column stats / row section with the same table encoded as THREE row shards, whose positions all stay
below 2^32); times the reference encoder on the host ("cpu_baseline"); and times the shipped convertDWfile
binary as a child process on a page-cache-warm 2 GB file ("e2e_cli": compressor child replaced by cat; as shipped, gzip
deflated in the process on several threads; and the reference's way, one gzip child) -- SURVEY.md 8d protocol (b); none of
these rates is ever `value`.  Other table shapes (--cols, --p-empty, --dead-cols, --id-every) are for measurements, not
valid bench lines.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--gb G] [--cols C] [--no-cpu] [--no-verify] [--no-e2e] [--scaling weak|strong]
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def cpu_baseline(desc, sample, cores_note=1):
    """Times the reference encoder (oracle/_ref, kind 'reference') or the C oracle ('port') on `sample`."""
    from oracle import oracle
    mb = len(sample) / 1e6
    ref_out = None
    if oracle.have_ref():
        d = tempfile.mkdtemp(prefix="zdwbase_")
        try:
            with open(os.path.join(d, "s.sql"), "wb") as f:
                f.write(sample)
            with open(os.path.join(d, "s.desc.sql"), "wb") as f:
                f.write(desc)
            env = dict(os.environ)
            env["PATH"] = oracle.REF_SHIM + os.pathsep + env.get("PATH", "")
            t0 = time.perf_counter()
            p = subprocess.run([oracle.REF_BIN, "-q", os.path.join(d, "s.sql")], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            out_bytes = os.path.getsize(os.path.join(d, "s.zdw.gz")) if p.returncode == 0 else 0
            if p.returncode == 0:
                with open(os.path.join(d, "s.zdw.gz"), "rb") as f:
                    ref_out = f.read()              # (the compressor child is `cat`: the raw stream) -- compared with the HIP encode of the same rows
            kind = "reference"
        finally:
            shutil.rmtree(d, ignore_errors=True)
        if p.returncode != 0:
            return {"value": None, "unit": "MB/s", "cores": 1, "kind": kind, "sample": "reference failed: %s" % p.stderr[-200:]}
    else:
        t0 = time.perf_counter()
        rc, out, _ = oracle.encode(desc, sample)
        dt = time.perf_counter() - t0
        out_bytes = len(out) if rc == 0 else 0
        ref_out = out if rc == 0 else None
        kind = "port"
    return {"_out": ref_out, "value": round(mb / dt, 2), "unit": "MB/s", "cores": 1, "kind": kind,
            "sample": "first %.1f MB (whole rows) of the same synthetic table, compressor child replaced by cat, %.1f s, ratio %.2f"
                      % (mb, dt, len(sample) / max(1, out_bytes))}


def pmc_traffic(kernel, nbytes):
    """HBM bytes per launch of `kernel` from the latest committed rocprofv3 --pmc summary (profiles/), collected
    in separate FETCH_SIZE / WRITE_SIZE passes of this same command and corrected as MI355X_MICROARCH.md says;
    None when no summary matches this workload size."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        hit = [k for k in d.get("kernels", {}) if kernel and kernel in k]
        if abs(d.get("input_bytes", 0) - nbytes) <= 0.02 * nbytes and hit:
            best = (d["kernels"][hit[0]]["hbm_bytes_per_launch"], os.path.basename(f))
    return best if best else (None, None)


def verify_outputs(torch, capi, sharded, table, buf, nbytes, enc, nrows, p1, dict_ptr_len, rows_ptr_len, col_stats):
    """Size-independent properties of the block the timed loop produced last (outputs still in HBM), checked after
    the timed region.  Raises AssertionError with what is wrong; returns a dict of what was checked."""
    import hashlib
    import numpy as np
    dev = buf.device
    dptr, dlen = dict_ptr_len
    rptr, rlen = rows_ptr_len
    col_size, col_base = col_stats
    assert p1.num_rows == nrows and p1.is_last_block == 1, "row count %d != %d" % (p1.num_rows, nrows)
    sec = sharded._wrap(dptr, dlen, "|u1", str(dev)).cpu().numpy()
    w = int(sec[0])
    size = int.from_bytes(sec[1:1 + w].tobytes(), "little")
    blob = sec[1 + w:]
    assert blob.size == size and blob[0] == 0 and blob[-1] == 0, "dictionary header says %d bytes, section holds %d" % (size, blob.size)
    entries = blob[1:].tobytes().split(b"\0")[:-1]
    assert len(entries) == p1.dict_entries, "dictionary holds %d strings, pass 1 counted %d" % (len(entries), p1.dict_entries)
    assert all(a < b for a, b in zip(entries, entries[1:])), "dictionary is not in strict strcmp order"
    del entries
    used = [c for c in range(len(col_size)) if col_size[c]]
    assert len(col_base) == len(used)
    nflag = (len(used) + 7) // 8
    sizes = np.array([col_size[c] for c in used], dtype=np.int64)
    rows = sharded._wrap(rptr, rlen, "|u1", str(dev))
    head = rows[:min(rlen, 64 << 20)].cpu().numpy()
    pos = 0
    walked = 0
    while walked < min(nrows, 20000) and pos + nflag <= head.size:
        flags = np.unpackbits(head[pos:pos + nflag], bitorder="little")[:len(used)].astype(bool)
        pos += nflag + int(sizes[flags].sum())
        walked += 1
    assert pos <= rlen and (walked < nrows or pos == rlen), "row walk left the section"
    h_dict = hashlib.sha256(sec.tobytes()).hexdigest()
    h_rows = hashlib.sha256()
    for o in range(0, rlen, 256 << 20):
        h_rows.update(rows[o:o + (256 << 20)].cpu().numpy().tobytes())
    h_rows = h_rows.hexdigest()
    # the same rows as three shards of one block (every shard's positions and offsets stay below 2^32)
    cuts = [0]
    for i in (1, 2):
        t = nbytes * i // 3
        win = buf[t:t + (1 << 20)]
        cuts.append(t + int((win == 10).nonzero()[0][0].item()) + 1)
    cuts.append(nbytes)
    encs = []
    for i in range(3):
        part = buf[cuts[i]:cuts[i + 1]]
        if part.data_ptr() % 16:
            part = part.clone()
        e = capi.Encoder(table.types, device=dev.index or 0)
        e.set_input_device(part.data_ptr(), part.numel(), keepalive=part)
        encs.append(e)
    torch.cuda.synchronize()
    blocks = sharded.encode_shards_partitioned_local(encs, device=str(dev))
    got_dict = hashlib.sha256(b"".join(b.dict_section for b in blocks)).hexdigest()
    hr = hashlib.sha256()
    for b in blocks:
        hr.update(b.rows)
        assert b.col_size == col_size and b.col_base == col_base, "column stats differ between the sharded and the single block"
    assert sum(b.num_rows for b in blocks) == nrows
    assert got_dict == h_dict, "dictionary section differs between 3 shards and 1"
    assert hr.hexdigest() == h_rows, "row section differs between 3 shards and 1"
    for e in encs:
        e.close()
    return {"rows": int(nrows), "dict_entries": int(p1.dict_entries), "rows_walked": walked, "dict_sha256": h_dict[:16], "rows_sha256": h_rows[:16],
            "checks": "row count; dictionary size header, entry count and strict strcmp order of every entry; walk of the first rows; "
                      "dictionary, column stats and row bytes equal to the same table encoded as 3 row shards"}


def reference_prefix_equal(capi, table, sample, ref_file, device):
    """The bytes the CPU baseline wrote for `sample` (the head of the benchmark's table) against the HIP encode of the same rows:
    the whole block -- header, dictionary, column stats, rows -- byte for byte (the file header is the host shim's)."""
    import hashlib
    names = [ln.split(b"\t")[0] for ln in table.desc.splitlines() if ln]
    header_len = 2 + 4 + sum(len(n) + 1 for n in names) + 1 + 3 * len(names)        # version, metadata length, names, types, char sizes
    enc = capi.Encoder(table.types, device=device)
    try:
        enc.set_input_host(sample)
        rc, blk, p1 = enc.encode_block()
        got = blk.tobytes() if rc == 0 and blk is not None else b""
    finally:
        enc.close()
    want = ref_file[header_len:]
    return {"equal": got == want, "bytes_compared": len(want), "hip_sha256": hashlib.sha256(got).hexdigest()[:16],
            "reference_sha256": hashlib.sha256(want).hexdigest()[:16], "rows": int(p1.num_rows)}


def e2e_cli(torch, table, gb=2.0):
    """SURVEY.md 8d protocol (b): the shipped convertDWfile binary as a child process on a page-cache-warm file,
    wall clock from exec to exit (process start, HIP start-up, file read, H2D, encode, D2H, pipe to the compressor
    child, rename): compressor child replaced by `cat`, and the default gzip."""
    import stat
    from zdw_amd import build
    rows = table.rows_for_bytes(gb * 1e9)
    buf, nbytes = table.generate(0, rows)
    d = tempfile.mkdtemp(prefix="zdwcli_")
    res = {"bytes": int(nbytes), "file": "synthetic table of the same shape, page-cache warm"}
    try:
        with open(os.path.join(d, "t.sql"), "wb") as f:
            f.write(buf[:nbytes].cpu().numpy().tobytes())
        with open(os.path.join(d, "t.desc.sql"), "wb") as f:
            f.write(table.desc)
        del buf
        shim = os.path.join(d, "shim")
        os.makedirs(shim)
        sp = os.path.join(shim, "gzip")
        with open(sp, "w") as f:
            f.write("#!/bin/sh\nexec cat\n")
        os.chmod(sp, os.stat(sp).st_mode | stat.S_IXUSR)
        with open(os.path.join(d, "t.sql"), "rb") as f:          # warm the page cache
            while f.read(64 << 20):
                pass
        # cat_shim: the compressor child replaced by cat; gzip: the command line as shipped (deflate in the process on several
        # threads); gzip_child: the reference's way, one `gzip` child (ZDW_GZIP_THREADS=0)
        plain = os.environ.get("PATH", "")
        for label, path, threads in (("cat_shim", shim + os.pathsep + plain, "0"), ("gzip", plain, None), ("gzip_child", plain, "0")):
            env = dict(os.environ, PATH=path)
            env.pop("ZDW_GZIP_THREADS", None)
            if threads is not None:
                env["ZDW_GZIP_THREADS"] = threads
            t0 = time.perf_counter()
            r = subprocess.run([build.CLI_EXE, "-q", os.path.join(d, "t.sql")], env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            dt = time.perf_counter() - t0
            outp = os.path.join(d, "t.zdw.gz")
            if r.returncode != 0 or not os.path.exists(outp):
                res[label + "_MBps"] = None
                res[label + "_error"] = r.stderr[-200:].decode(errors="replace")
                continue
            res[label + "_MBps"] = round(nbytes / 1e6 / dt, 1)
            res[label + "_s"] = round(dt, 2)
            if label == "gzip":
                res["gzip_threads"] = min(os.cpu_count() or 1, 16)
                res["gzip_bytes"] = os.path.getsize(outp)
                res["gzip_reads_back"] = subprocess.run(["gzip", "-t", outp]).returncode == 0
            if label == "cat_shim":
                # the other direction: that raw stream back to text on the device (zdw_dec_*), the second of two
                # runs; checked against the file it came from on its first rows, by row count and by length
                try:
                    res["decode"] = decode_leg(table, open(outp, "rb").read(), os.path.join(d, "t.sql"), rows)
                except Exception as e:  # noqa: BLE001
                    res["decode"] = {"error": repr(e)[:200]}
            os.unlink(outp)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return res


def decode_leg(table, z, sql_path, rows):
    from zdw_amd import capi
    dec = capi.Decoder(0)
    try:
        out = None
        for run in range(2):
            assert dec.set_input(z) == 0, dec.error()
            stages, text_bytes, nrows, first = {}, 0, 0, None
            while True:
                rc, blk = dec.next_block()
                assert rc == 0, dec.error()
                if blk.num_rows:
                    nrows += blk.num_rows
                    text_bytes += blk.text_bytes
                    for k, v in dec.stage_ms().items():
                        stages[k] = round(stages.get(k, 0.0) + v, 3)
                    if first is None and run == 1:
                        first = dec.text()[:8 << 20]
                if blk.is_last_block:
                    break
            dev_ms = sum(stages.values())
            out = {"zdw_bytes": len(z), "text_bytes": int(text_bytes), "device_ms": round(dev_ms, 2), "text_MBps": round(text_bytes / 1e3 / dev_ms, 1) if dev_ms else None,
                   "stage_ms": stages}
        integer = [t in (7, 8, 9, 10, 12, 13, 14, 15) for t in table.types]
        first = first[:first.rfind(b"\n") + 1]
        k = first.count(b"\n")
        want = bytearray()
        with open(sql_path, "rb") as f:
            for _ in range(k):
                fs = f.readline()[:-1].split(b"\t")
                want += b"\t".join((x or b"0") if integer[i] else (x or b"0.000000000000") if table.types[i] == 11 else x for i, x in enumerate(fs)) + b"\n"
        out["verified"] = bool(nrows == rows and bytes(want) == first)
        out["checks"] = "row count; the first %d rows equal the input's (empty integer / decimal fields read 0 / 0.000000000000)" % k
        return out
    finally:
        dec.close()


def stream_copy_gbps(torch, src):
    """Device-to-device copy of (up to) 4 GB of the input image: bytes read + bytes written per second."""
    n = min(int(src.numel()), 4 << 30)
    dst = torch.empty(n, dtype=src.dtype, device=src.device)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    dst.copy_(src[:n])
    best = 0.0
    for _ in range(5):
        a.record()
        dst.copy_(src[:n])
        b.record()
        b.synchronize()
        best = max(best, 2.0 * n / (a.elapsed_time(b) * 1e-3) / 1e9)
    del dst
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--gb", type=float, default=10.0, help="input GB (1e9 bytes) per GPU")
    ap.add_argument("--cols", type=int, default=200)
    ap.add_argument("--cpu-mb", type=float, default=600.0, help="size of the CPU-baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only for rehearsals)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--word-maxlen", type=int, default=24, help="longest vocabulary word (24 = BASELINE's table; more: not a valid bench line)")
    ap.add_argument("--independent-blocks", action="store_true",
                    help="N > 1: every rank writes its rows as its own ZDW block (own dictionary, no exchange; a different, equally valid file: "
                         "SURVEY 8f-3) instead of one block row-sharded over the ranks")
    ap.add_argument("--p-empty", type=float, default=0.25, help="fraction of empty fields (0.25 = BASELINE's table; other values: not a valid bench line)")
    ap.add_argument("--dead-cols", type=float, default=0.0, help="fraction of columns empty in every row (0 = BASELINE's table)")
    ap.add_argument("--id-every", type=int, default=0, help="every k-th column holds practically unique 24-character ids (C5's table: --cols 1000 --id-every 20; 0 = BASELINE's table)")
    ap.add_argument("--pass1-only", action="store_true", help="profiling: time pass 1 alone (not a valid bench line)")
    ap.add_argument("--no-verify", action="store_true", help="skip the property checks of the outputs after the timed region")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end timing of the convertDWfile binary after the timed region")
    ap.add_argument("--e2e-gb", type=float, default=2.0, help="size of the file the convertDWfile binary is timed on")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="N > 1: weak = every rank holds its own --gb of rows (default); strong = ONE table of --gb split over the ranks "
                         "(BASELINE configs[3] as written).  With N > 1 the line of a weak run also carries a strong measurement.")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from zdw_amd import capi, sharded, synth

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0 if args.one_device else int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local)
    force_dist = world == 1 and os.environ.get("ZDW_BENCH_FORCE_DIST") == "1"      # measurements: the N > 1 driver with one rank (not a valid bench line)
    if force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or force_dist:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)

    # ---- synthetic table, generated in HBM; rank r holds rows [r*R, (r+1)*R) ----
    table = synth.DeviceTable(ncols=args.cols, device="cuda:%d" % local, word_maxlen=args.word_maxlen, p_empty=args.p_empty, dead_cols=args.dead_cols, id_every=args.id_every)
    rows_total = table.rows_for_bytes(args.gb * 1e9)

    def rank_rows(mode):
        """(first row, rows) of this rank: weak = its own --gb; strong = its share of ONE table of --gb"""
        if mode == "weak" or world == 1:
            return rank * rows_total, rows_total
        lo_r, hi_r = rows_total * rank // world, rows_total * (rank + 1) // world
        return lo_r, hi_r - lo_r

    row0, rows_per_rank = rank_rows(args.scaling)
    buf, nbytes = table.generate(row0, rows_per_rank)
    torch.cuda.synchronize()

    # N > 1: every rank's table must hold the union of all ranks' strings (it is not regrown after pass 1):
    # 2^25 slots of 32 B = 1 GiB take 21 M distinct strings (up to 4 ranks of this table), 2^26 take 42 M
    merge = os.environ.get("ZDW_SHARD_MERGE", "partition")
    slots = 0                     # the range-partitioned dictionary keeps a rank's table to its own strings
    if world > 1 and merge == "allgather":
        slots = (1 << 22) if args.gb < 2 else ((1 << 25) if world <= 4 else (1 << 26))
    enc = capi.Encoder(table.types, device=local, dict_slots=slots)
    enc.set_input_device(buf.data_ptr(), nbytes, keepalive=buf)
    enc.enable_timing(True)

    last = {}

    def step():
        if world > 1 and args.independent_blocks:
            blocks = sharded.encode_independent_blocks_dist(enc, rank, world, fetch=False)
            return sum(b[0] for b in blocks), sum(b[1] for b in blocks), sum(b[2] for b in blocks)
        if world > 1 or force_dist:
            return sharded.encode_shard_dist(enc, rank, world, merge=merge)
        rc, p1 = enc.pass1()
        if rc:
            raise capi.ZdwError(rc, "pass1")
        if args.pass1_only:
            return p1.num_rows, 0, 0
        dptr, dlen, _ = enc.dict_section_device()
        stats = enc.stats()
        rptr, rlen = enc.pass2_device()
        last.update(p1=p1, dict=(dptr, dlen), rows=(rptr, rlen), stats=stats)
        return p1.num_rows, dlen, rlen

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    enc.enable_timing(True)      # reset stage accumulators after warm-up
    fence()
    t0 = time.perf_counter()
    step_ends = []
    for _ in range(args.steps):
        nrows, dict_len, rows_len = step()
        step_ends.append(time.perf_counter())      # a step returns once its lengths are on the host: no extra sync
    fence()
    dt = time.perf_counter() - t0
    step_ms = sorted((b - a) * 1e3 for a, b in zip([t0] + step_ends[:-1], step_ends))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([nbytes, dict_len + rows_len + nrows * 0], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        total_in, total_out = int(tot[0].item()), int(tot[1].item())
    else:
        total_in, total_out = nbytes, dict_len + rows_len
    ms = dt / args.steps * 1e3
    stages = enc.stage_times()

    # N > 1, weak run: the same ranks on ONE table of --gb (BASELINE configs[3] as written), reported beside the weak number
    strong = None
    if world > 1 and args.scaling == "weak" and not args.independent_blocks:
        s_row0, s_rows = rank_rows("strong")
        del buf
        enc.close()
        buf, s_bytes = table.generate(s_row0, s_rows)
        torch.cuda.synchronize()
        enc = capi.Encoder(table.types, device=local, dict_slots=slots)
        enc.set_input_device(buf.data_ptr(), s_bytes, keepalive=buf)
        for _ in range(max(1, args.warmup)):
            step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        sdt = time.perf_counter() - t1
        t = torch.tensor([sdt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        tb = torch.tensor([s_bytes], dtype=torch.int64, device="cuda")
        dist.all_reduce(tb)
        strong = {"value": round(int(tb.item()) / 1e6 / (float(t.item()) / args.steps), 1), "unit": "MB/s", "ms_per_step": round(float(t.item()) / args.steps * 1e3, 3),
                  "input_bytes_total": int(tb.item()), "workload": "ONE synthetic table of %.1f GB split over %d ranks by rows" % (int(tb.item()) / 1e9, world)}

    if rank == 0:
        out_local = dict_len + rows_len + 9 + args.cols * 9
        per = {k: round(v[0] / max(1, args.steps), 3) for k, v in stages.items() if v[1]}
        # dominant kernel: pass 1 (tokenise + dedupe + parse).  Algorithmic bytes per launch: one read of
        # the block's TSV (B_in); the whole encode is priced at 2*B_in + B_out.
        dom = max((k for k in per if k in ("scan_rows", "pass1", "pass2_emit")), key=lambda k: per[k])
        alg = {"scan_rows": nbytes, "pass1": nbytes, "pass2_size": 0, "pass2_emit": rows_len, "dict_sort": 0,
               "dict_compact": 0, "dict_emit": dict_len}.get(dom, 0)
        dom_ms = per[dom]
        achieved = alg / (dom_ms * 1e-3) / 1e9 if dom_ms else 0.0
        b_alg = 2 * nbytes + out_local
        traffic, traffic_note = pmc_traffic({"pass1": "zdw::k_pass1<true", "scan_rows": "zdw::k_count_row_starts",
                                             "pass2_emit": "zdw::k_pass2_emit_cols", "pass2_size": "zdw::k_pass2_size_cols"}.get(dom), nbytes)
        res = {
            "metric": "TSV->ZDW encode throughput (input bytes / wall time, input resident in HBM, raw .zdw left in HBM)",
            "value": round(total_in / 1e6 / (dt / args.steps), 1), "unit": "MB/s",
            "n_gpus": world, "comm_ranks": (dist.get_world_size() if (world > 1 or force_dist) else 1),      # ranks of the RCCL communicator the step ran on
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": args.scaling if world > 1 else "weak", "vs_baseline": None, "dtype": "u8/u64", "data": "synthetic",
            "config": {"workload": "synthetic TSV %.1f GB x %d cols %s, mixed VARCHAR/LONG, Zipf(1.1) strings over 2^20 words, 25%% empty, 30%% repeat (BASELINE configs[3])"
                                   % ((total_in if args.scaling == "strong" else nbytes) / 1e9, args.cols,
                                      "split over the GPUs by rows" if (args.scaling == "strong" and world > 1) else "per GPU"),
                       "rows_per_gpu": int(nrows), "input_bytes_per_gpu": int(nbytes), "zdw_bytes_per_gpu": int(out_local),
                       "compression_ratio": round(nbytes / max(1, out_local), 3),
                       "parallelism": ("one ZDW block per GPU, %d GPUs, independent dictionaries (not the single-block file)" % world)
                                      if (world > 1 and args.independent_blocks) else "one ZDW block row-sharded over %d GPU(s)" % world},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "algorithmic_bytes_per_launch": int(alg),
                         "whole_encode": {"b_alg": int(b_alg), "achieved": round(b_alg / (ms * 1e-3) / 1e9, 1),
                                          "frac": round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}},
            "stage_ms": per,
            "step_ms": {"median": round(step_ms[len(step_ms) // 2], 3), "min": round(step_ms[0], 3)},
        }
        if strong is not None:
            res["strong_scaling"] = strong
        # ---- after the timed region: are the bytes that were just timed the right ones? ----
        if world == 1 and not args.pass1_only and not args.no_verify and last:
            try:
                res["verify"] = verify_outputs(torch, capi, sharded, table, buf, nbytes, enc, rows_per_rank, last["p1"], last["dict"], last["rows"], last["stats"])
                res["verified"] = True
            except AssertionError as e:
                res["verified"] = False
                res["verify"] = {"error": str(e)}
        else:
            res["verified"] = None
        # what a plain streaming copy reaches on this box (quote the fraction of both peaks)
        copy_gbps = stream_copy_gbps(torch, buf)
        res["roofline"]["stream_copy"] = {"measured": round(copy_gbps, 1), "unit": "GB/s (read + write)",
                                          "frac_of_it": round(achieved / copy_gbps, 4) if copy_gbps else None}
        if not args.no_cpu and world == 1:
            # bounded sample: whole rows from the head of rank 0's table (N = 1 only: the other ranks would sit in the barrier)
            want = int(min(args.cpu_mb * 1e6, nbytes))
            head = buf[:want].cpu().numpy().tobytes()
            cut = head.rfind(b"\n") + 1
            res["cpu_baseline"] = cpu_baseline(table.desc, head[:cut])
            res["cpu_baseline"]["host_cores"] = os.cpu_count()
            # ... and its bytes against the HIP path's for the same rows: reference bytes at bench scale (checker only; outside the timed region)
            ref_file = res["cpu_baseline"].pop("_out", None)
            if ref_file is not None and not args.no_verify:
                try:
                    cmp_ = reference_prefix_equal(capi, table, head[:cut], ref_file, local)
                    res.setdefault("verify", {})["reference_prefix"] = cmp_
                    res["verify"]["reference_prefix_equal"] = cmp_["equal"]
                    if not cmp_["equal"]:
                        res["verified"] = False
                except Exception as e:  # noqa: BLE001
                    res.setdefault("verify", {})["reference_prefix"] = {"error": repr(e)[:200]}
                    res["verified"] = False
            del ref_file
        else:
            res["cpu_baseline"] = {"value": None, "unit": "MB/s", "cores": 1, "kind": "reference", "sample": "skipped (--no-cpu, or N > 1: measured at N = 1)"}
        if not args.no_e2e and not args.pass1_only and world == 1:
            enc.close()
            del buf
            torch.cuda.empty_cache()
            try:
                res["e2e_cli"] = e2e_cli(torch, table, args.e2e_gb)
            except Exception as e:  # noqa: BLE001
                res["e2e_cli"] = {"error": repr(e)[:200]}
        print(json.dumps(res))
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()
    enc.close()


if __name__ == "__main__":
    main()
