#!/usr/bin/env python3
"""BASELINE configs[1] (C2, movie_tickets: narrow schema, dictionary ~ rows) at a size that loads the GPU, and the compression ratios of
SURVEY.md 8d(ii) for C2 / C3 against test-files/README.md.

    python tools/run_c2.py --copies 128 --out gpurun_out/c2.json

The table: the reference's own movie_tickets.sql (tests/golden/movie_tickets.sql.xz, 524 160 rows x 6 columns, 32.65 MB; its date_time
column is unique per row, so the dictionary has as many entries as the file has rows -- the shape on which the reference spends
60-70 % of its time in the red-black tree), `copies` times in HBM, the year of every copy's date_time moved on by one so
that the strings stay unique across copies (--copies 128: 4.18 GB, 67 M rows, 67 M dictionary entries of 19 characters).
Timed: one block, pass 1 + dictionary + statistics + pass 2, outputs left in HBM (median of --steps).  Checked: row count, entry
count (copies x 524 160 date_times + what the other columns hold once), dictionary order on its first 64 MB, and the first copy's rows
byte for byte against the golden .zdw's row section where the dictionary offsets allow it (copies = 1 only: the whole file is the golden).
Ratios: the real C2 and C3 files through the HIP path, then `gzip -6` / `xz -6` of the raw stream on the host."""
import argparse
import hashlib
import json
import lzma
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def rd(name):
    p = os.path.join(G, name)
    return lzma.open(p).read() if name.endswith(".xz") else open(p, "rb").read()


def desc_types(desc):
    from tests import zdw_walk  # noqa: F401  (test helper only for the check of copies == 1)
    out = []
    for ln in desc.splitlines():
        t = ln.split(b"\t")[1].lower()
        if t.startswith(b"varchar"):
            out.append(0)
        elif t.startswith(b"datetime"):
            out.append(2)
        elif t.startswith(b"decimal"):
            out.append(11)
        elif t.startswith(b"smallint"):
            out.append(8 if b"unsigned" in t else 13)
        else:
            out.append(9 if b"unsigned" in t else 14)
    return out


def ratios(capi):
    """raw / gzip / xz sizes of the real C2 and C3 files encoded by the HIP path (format version 11: 4 bytes more than the README's v10 files)"""
    out = {}
    readme = {"movie_tickets": {"tsv": 32653800, "zdw": 18290691, "zdw_gz": 4365887, "zdw_xz": 2040472},
              "analytics-hits": {"tsv": 14468990, "zdw": 1303440, "zdw_gz": 312272, "zdw_xz": 227456}}
    import shutil
    from tests import parity_util as pu
    gzip_exe, xz_exe = shutil.which("gzip"), shutil.which("xz")          # (before the sandbox puts its `gzip` = cat in front of PATH)
    sb = pu.CliSandbox()
    try:
        for name in ("movie_tickets", "analytics-hits"):
            tsv, desc = rd(name + ".sql.xz"), rd(name + ".desc.sql")
            rc, z = sb.run(desc, tsv, args=["-q"])
            assert rc == 0
            gz = subprocess.run([gzip_exe, "-6", "-c"], input=z, capture_output=True).stdout
            xz = subprocess.run([xz_exe, "-6", "-c"], input=z, capture_output=True).stdout
            r = readme[name]
            out[name] = {"tsv_bytes": len(tsv), "zdw_bytes": len(z), "zdw_gz_bytes": len(gz), "zdw_xz_bytes": len(xz),
                         "ratio_raw": round(len(tsv) / len(z), 3), "ratio_gz": round(len(tsv) / len(gz), 2), "ratio_xz": round(len(tsv) / len(xz), 2),
                         "readme": r, "readme_ratio_raw": round(r["tsv"] / r["zdw"], 3), "readme_ratio_gz": round(r["tsv"] / r["zdw_gz"], 2),
                         "readme_ratio_xz": round(r["tsv"] / r["zdw_xz"], 2),
                         "zdw_sha256": hashlib.sha256(z).hexdigest()[:16]}
    finally:
        sb.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--copies", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--out", default="gpurun_out/c2.json")
    ap.add_argument("--no-ratios", action="store_true")
    ap.add_argument("--pass1-only", action="store_true", help="time pass 1 alone, check nothing (ablations with the experiments build: ZDW_DEBUG_SKIP)")
    args = ap.parse_args()
    import numpy as np
    import torch
    from zdw_amd import capi

    base = rd("movie_tickets.sql.xz")
    desc = rd("movie_tickets.desc.sql")
    types = desc_types(desc)
    nb = len(base)
    b = torch.frombuffer(bytearray(base), dtype=torch.uint8).to("cuda:0")
    starts = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda:0"), (b == 10).nonzero().flatten()[:-1] + 1])
    rows1 = int(starts.numel())
    buf = b.repeat(args.copies)
    for k in range(1, args.copies):          # the year of copy k: 2018 + k
        y = ("%04d" % (2018 + k)).encode()
        for j in range(4):
            buf[k * nb + starts + j] = y[j]
    n = buf.numel()
    torch.cuda.synchronize()
    rows = rows1 * args.copies
    enc = capi.Encoder(types, device=0)
    enc.set_input_device(buf.data_ptr(), n, keepalive=buf)
    if args.pass1_only:
        ms = []
        for step in range(args.steps + 1):
            enc.set_input_device(buf.data_ptr(), n, keepalive=buf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc, p1 = enc.pass1()
            torch.cuda.synchronize()
            if step:
                ms.append(round((time.perf_counter() - t0) * 1e3, 3))
        print(json.dumps({"pass1_only_ms": ms, "rc": rc, "rows": int(p1.num_rows), "dict_entries": int(p1.dict_entries)}))
        return
    times, stages = [], None
    for step in range(args.steps + 1):
        if step == 1:
            enc.enable_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc, p1 = enc.pass1()
        assert rc == 0, rc
        dptr, dlen, isz = enc.dict_section_device()
        cs, cb = enc.stats()
        rptr, rlen = enc.pass2_device()
        torch.cuda.synchronize()
        if step:
            times.append((time.perf_counter() - t0) * 1e3)
    stages = {k: round(v[0] / args.steps, 3) for k, v in enc.stage_times().items() if v[1]}
    assert p1.num_rows == rows and p1.is_last_block == 1
    from zdw_amd import sharded
    sec = sharded._wrap(dptr, dlen, "|u1", "cuda:0")
    assert int((sec[1 + isz:] == 0).sum().item()) == p1.dict_entries + 1
    head = sec[1 + isz + 1:1 + isz + 1 + (64 << 20)].cpu().numpy().tobytes().split(b"\0")[:-1]
    assert all(x < y for x, y in zip(head, head[1:])), "dictionary order"
    ms = sorted(times)[len(times) // 2]
    res = {"workload": "C2 shape: movie_tickets.sql x %d copies (date_time unique per row), %.2f GB, %d rows x 6 columns" % (args.copies, n / 1e9, rows),
           "input_bytes": int(n), "rows": int(rows), "dict_entries": int(p1.dict_entries), "dict_bytes": int(dlen), "row_bytes": int(rlen),
           "ratio_raw": round(n / (dlen + rlen + 9 + 6 + 48), 3), "encode_ms": round(ms, 2), "input_GBps": round(n / 1e6 / ms, 1),
           "stage_ms": stages, "sort_share": round(stages.get("dict_sort", 0.0) / ms, 3),
           "dict_entries_order_checked": len(head)}
    enc.close()
    del buf
    if not args.no_ratios:
        res["compression"] = ratios(capi)
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
