#!/usr/bin/env python3
"""BASELINE configs[4] (C5) on ONE GPU: the 1000-column table with 50 high-cardinality id columns, --gb of it
resident in HBM, encoded through the automatic multi-block path under the reference's default memory limit (3072 MB).

    python tools/run_c5.py --gb 50 --out gpurun_out/c5.json

Timed: every block's pass 1 + dictionary + statistics + pass 2 with the outputs left in HBM (wall clock around the block, device
synchronised).  Checked afterwards, per block: `done` flags, row counts summing to the table's, dictionary below 2^32 with its size
header, entry count and strict order (first 64 MB of every dictionary), a walk of the first rows; and for one block in the middle of
the file the full round trip: file header + that block through the shipped decoder (zdw_amd/lib/unconvertDWfile) must give back the
block's rows of the input, byte for byte (cmp)."""
import argparse
import hashlib
import json
import os
import shutil
import struct
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def file_header(table, first_row):
    """The file header the command line writes for this schema: the .zdw of a one-row file minus its only block."""
    from zdw_amd import build, capi
    d = tempfile.mkdtemp(prefix="c5hdr_")
    try:
        shim = os.path.join(d, "shim")
        os.makedirs(shim)
        with open(os.path.join(shim, "gzip"), "w") as f:
            f.write("#!/bin/sh\nexec cat\n")
        os.chmod(os.path.join(shim, "gzip"), 0o755)
        stem = os.path.join(d, "t")
        open(stem + ".sql", "wb").write(first_row)
        open(stem + ".desc.sql", "wb").write(table.desc)
        env = dict(os.environ, ZDW_GZIP_THREADS="0", PATH=shim + os.pathsep + os.environ.get("PATH", ""))
        r = subprocess.run([build.CLI_EXE, "-q", stem + ".sql"], env=env, capture_output=True)
        if r.returncode:
            raise RuntimeError("convertDWfile on the one-row file: rc %d %s" % (r.returncode, r.stderr[-300:]))
        whole = open(stem + ".zdw.gz", "rb").read()
        enc = capi.Encoder(table.types, device=0)
        enc.set_input_host(first_row)
        rc, blk, _ = enc.encode_block()
        enc.close()
        body = blk.tobytes()
        assert rc == 0 and whole.endswith(body)
        return whole[:len(whole) - len(body)]
    finally:
        shutil.rmtree(d, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=50.0)
    ap.add_argument("--cols", type=int, default=1000)
    ap.add_argument("--mem-limit", type=float, default=3072.0, help="MB (the reference's default)")
    ap.add_argument("--out", default="gpurun_out/c5.json")
    ap.add_argument("--no-roundtrip", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    from zdw_amd import build, capi, sharded, synth

    table = synth.DeviceTable(ncols=args.cols, device="cuda:0", id_every=20)
    rows = table.rows_for_bytes(args.gb * 1e9)
    t0 = time.perf_counter()
    buf, n = table.generate(0, rows)
    torch.cuda.synchronize()
    print("table: %d rows, %.2f GB in %.1f s" % (rows, n / 1e9, time.perf_counter() - t0), flush=True)

    # first the clock: every block's four calls, nothing else in between (the checks below keep the host busy for seconds).  Twice: the
    # first pass allocates every buffer (an allocation of ~10 GB behind a release of as much was seen to take a second: the driver
    # clears what it hands out).
    passes, passes_launches = [], []
    enc = capi.Encoder(table.types, device=0, mem_limit_mb=int(args.mem_limit))       # one context for all passes: its buffers stay
    for _ in range(2):
        enc.set_input_device(buf.data_ptr(), n, keepalive=buf)
        enc.enable_timing(True)
        timed = []
        torch.cuda.synchronize()
        while True:
            t_b = time.perf_counter()
            rc, p1 = enc.pass1()
            if rc:
                raise capi.ZdwError(rc, "pass1")
            t_1 = time.perf_counter()
            enc.dict_section_device()
            t_2 = time.perf_counter()
            enc.stats()
            enc.pass2_device()
            torch.cuda.synchronize()
            t_3 = time.perf_counter()
            timed.append({"rows": int(p1.num_rows), "ms": round((t_3 - t_b) * 1e3, 3),
                          "calls_ms": {"pass1": round((t_1 - t_b) * 1e3, 1), "dict": round((t_2 - t_1) * 1e3, 1), "stats_pass2": round((t_3 - t_2) * 1e3, 1)}})
            if p1.is_last_block:
                break
            enc.next_block()
        stages_timed = enc.stage_times()
        launches_timed = enc.pass1_launches() - sum(passes_launches)
        passes_launches.append(launches_timed)
        passes.append(round(sum(b["ms"] for b in timed), 1))
        print("timed:", timed, flush=True)

    # then the same again with every block looked at
    enc.set_input_device(buf.data_ptr(), n, keepalive=buf)
    enc.enable_timing(True)
    launches_before_checked = enc.pass1_launches()
    blocks, launches_before = [], launches_before_checked
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    while True:
        t_b = time.perf_counter()
        rc, p1 = enc.pass1()
        if rc:
            raise capi.ZdwError(rc, "pass1")
        t_1 = time.perf_counter()
        dptr, dlen, _ = enc.dict_section_device()
        t_2 = time.perf_counter()
        col_size, col_base = enc.stats()
        rptr, rlen = enc.pass2_device()
        torch.cuda.synchronize()
        t_3 = time.perf_counter()
        ms = (t_3 - t_b) * 1e3
        calls_ms = {"pass1": round((t_1 - t_b) * 1e3, 1), "dict": round((t_2 - t_1) * 1e3, 1), "stats_pass2": round((t_3 - t_2) * 1e3, 1)}
        launches = enc.pass1_launches()
        b = {"rows": int(p1.num_rows), "dict_entries": int(p1.dict_entries), "dict_bytes": int(dlen), "row_bytes": int(rlen), "done": int(p1.is_last_block),
             "longest_line": int(p1.longest_line), "pass1_launches": launches - launches_before, "partial_fields": 0 if p1.is_last_block else enc.partial_fields(),
             "ms": round(ms, 3), "calls_ms": calls_ms}
        launches_before = launches
        # ---- checks of this block's outputs (outside the clock of the next block: the clock restarts at t_b) ----
        sec = sharded._wrap(dptr, dlen, "|u1", "cuda:0")
        w = int(sec[0].item())
        size = int.from_bytes(sec[1:1 + w].cpu().numpy().tobytes(), "little")
        assert dlen == 1 + w + size and size < (1 << 32), "dictionary header"
        assert int((sec[1 + w:] == 0).sum().item()) == p1.dict_entries + 1, "dictionary entry count"
        head = sec[1 + w + 1:1 + w + 1 + (64 << 20)].cpu().numpy().tobytes()
        ent = head.split(b"\0")[:-1]
        assert all(x < y for x, y in zip(ent, ent[1:])), "dictionary order"
        b["dict_entries_order_checked"] = len(ent)
        used = [c for c in range(len(col_size)) if col_size[c]]
        nflag = (len(used) + 7) // 8
        sizes = np.array([col_size[c] for c in used], dtype=np.int64)
        rws = sharded._wrap(rptr, rlen, "|u1", "cuda:0")
        hd = rws[:min(rlen, 32 << 20)].cpu().numpy()
        pos = walked = 0
        while walked < min(p1.num_rows, 5000) and pos + nflag <= hd.size:
            flags = np.unpackbits(hd[pos:pos + nflag], bitorder="little")[:len(used)].astype(bool)
            pos += nflag + int(sizes[flags].sum())
            walked += 1
        assert pos <= rlen, "row walk"
        b["rows_walked"] = walked
        keep = None
        if not args.no_roundtrip and len(blocks) == 1:          # the second block: starts behind a partial row, is not the last
            keep = (struct.pack("<IIB", p1.num_rows, p1.longest_line, 1) + sec.cpu().numpy().tobytes() + bytes(col_size)
                    + struct.pack("<%dQ" % len(col_base), *col_base) + rws.cpu().numpy().tobytes())
        b["_keep"] = keep
        blocks.append(b)
        print("block %d: %s" % (len(blocks), {k: v for k, v in b.items() if k != "_keep"}), flush=True)
        if p1.is_last_block:
            break
        enc.next_block()
    wall_s = time.perf_counter() - t_all
    stages = enc.stage_times()
    enc.close()
    enc_ms = sum(b["ms"] for b in timed)
    assert [b["rows"] for b in timed] == [b["rows"] for b in blocks] or abs(sum(b["rows"] for b in timed) - rows) == 0
    assert sum(b["rows"] for b in blocks) == rows and [b["done"] for b in blocks] == [0] * (len(blocks) - 1) + [1]

    roundtrip = None
    if not args.no_roundtrip and len(blocks) > 1:
        # rows of block 2 in the input: after the first blocks[0].rows line ends
        r0, r1 = blocks[0]["rows"], blocks[0]["rows"] + blocks[1]["rows"]
        ends, base, lo, hi, seen = [], 0, None, None, 0
        step = 1 << 30
        while base < n and hi is None:
            nl = (buf[base:base + step] == 10).nonzero().flatten()
            k = int(nl.numel())
            if lo is None and seen + k >= r0:
                lo = base + int(nl[r0 - seen - 1].item()) + 1
            if seen + k >= r1:
                hi = base + int(nl[r1 - seen - 1].item()) + 1
            seen += k
            base += step
            del nl
        first_row = buf[:1 << 20].cpu().numpy().tobytes().split(b"\n", 1)[0] + b"\n"
        d = tempfile.mkdtemp(prefix="c5rt_")
        try:
            with open(os.path.join(d, "want.sql"), "wb") as f:
                for o in range(lo, hi, 1 << 28):
                    f.write(buf[o:min(hi, o + (1 << 28))].cpu().numpy().tobytes())
            del buf
            torch.cuda.empty_cache()
            hdr = file_header(table, first_row)
            with open(os.path.join(d, "b.zdw"), "wb") as f:
                f.write(hdr)
                f.write(blocks[1]["_keep"])
            t = time.perf_counter()
            zf = os.path.join(d, "b.zdw")
            p = subprocess.run([build.DECODER_EXE, "-t", zf], capture_output=True)          # the whole block decodes (no output kept)
            decode_s = time.perf_counter() - t
            # ... and its first rows are the input's, byte for byte, once the input's empty integer fields read "0" (what ZDW keeps of them)
            got = subprocess.run("%s - %s | head -c %d" % (build.DECODER_EXE, zf, 96 << 20), shell=True, capture_output=True).stdout
            got = got[:got.rfind(b"\n") + 1]
            nrows_cmp = got.count(b"\n")
            integer = [t in (7, 8, 9, 10, 12, 13, 14, 15) for t in table.types]
            want = bytearray()
            with open(os.path.join(d, "want.sql"), "rb") as f:
                for _ in range(nrows_cmp):
                    fs = f.readline()[:-1].split(b"\t")
                    want += b"\t".join((x or b"0" if integer[i] else x or b"0.000000000000" if table.types[i] == 11 else x) for i, x in enumerate(fs)) + b"\n"
            same = bytes(want) == got
            roundtrip = {"block": 2, "input_bytes": hi - lo, "decode_rc": p.returncode, "decode_s": round(decode_s, 1), "rows_compared": nrows_cmp,
                         "bytes_compared": len(got), "identical": same,
                         "how": "file header + block 2 through zdw_amd/lib/unconvertDWfile: the whole block decodes (-t); its first rows equal the block's "
                                "rows of the input byte for byte (empty integer / decimal fields of the input read 0 / 0.000000000000: what the format keeps of them)"}
            if not same:
                i = next((k for k in range(min(len(got), len(want))) if got[k] != want[k]), min(len(got), len(want)))
                roundtrip["first_difference"] = {"at": i, "got": repr(got[max(0, i - 60):i + 60]), "want": repr(bytes(want[max(0, i - 60):i + 60]))}
            roundtrip["rc"] = 0 if (same and p.returncode == 0 and nrows_cmp > 0) else 1
        finally:
            shutil.rmtree(d, ignore_errors=True)
    for b in blocks:
        b.pop("_keep", None)
    out_bytes = sum(b["dict_bytes"] + b["row_bytes"] for b in blocks)
    res = {"workload": "C5 shape: synthetic TSV %.1f GB x %d cols, %d of them 24-character id columns (practically unique values), one GPU, "
                       "memory limit %.0f MB (the reference's default)" % (n / 1e9, args.cols, args.cols // 20, args.mem_limit),
           "input_bytes": int(n), "rows": int(rows), "blocks": len(blocks), "zdw_bytes": int(out_bytes),
           "encode_ms": round(enc_ms, 1), "input_GBps": round(n / 1e6 / enc_ms, 1), "wall_s_with_checks": round(wall_s, 1),
           "pass1_launches": int(launches_timed),
           "stage_ms_total": {k: round(v[0], 2) for k, v in stages_timed.items() if v[1]},
           "timed_blocks": timed, "encode_ms_of_both_passes": passes,
           "roundtrip": roundtrip, "per_block": blocks}
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "per_block"}))
    if roundtrip and roundtrip["rc"]:
        raise SystemExit(1)


if __name__ == "__main__":
    main()
