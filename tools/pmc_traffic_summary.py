#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/profile_round.sh) into the
per-kernel HBM traffic JSON that bench.py reads from profiles/.

    python tools/pmc_traffic_summary.py gpurun_out/prof_<tag> profiles/r01_v2_pmc_hbm_traffic.json <input_bytes>

Counter unit: KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at
64 bytes, i.e. half the bytes of a streaming read -- reads are doubled; WRITE_SIZE is taken as is."""
import collections
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(dirname + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                k = r["Kernel_Name"].split("(")[0]
                tot[k] += float(r["Counter_Value"])
                n[k] += 1
    return tot, n


def main():
    src, dst, nbytes = sys.argv[1], sys.argv[2], int(sys.argv[3])
    fetch, nf = per_kernel(src + "/pmc_fetch", "FETCH_SIZE")
    write, nw = per_kernel(src + "/pmc_write", "WRITE_SIZE")
    out = {"input_bytes": nbytes, "kernels": {}, "note": __doc__.split("\n\n", 2)[2].strip().replace("\n", " ")}
    for k in sorted(set(fetch) | set(write)):
        if "zdw::" not in k:
            continue
        launches = max(nf.get(k, 0), nw.get(k, 0), 1)
        fr = fetch.get(k, 0.0) * 1024 / max(1, nf.get(k, 1))
        wr = write.get(k, 0.0) * 1024 / max(1, nw.get(k, 1))
        out["kernels"][k] = {"launches_profiled": launches, "fetch_raw_bytes_per_launch": int(fr), "fetch_corrected_bytes_per_launch": int(2 * fr),
                             "write_bytes_per_launch": int(wr), "hbm_bytes_per_launch": int(2 * fr + wr)}
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    for k, v in out["kernels"].items():
        if v["hbm_bytes_per_launch"] > 1e8:
            print("%-40s %8.2f GB" % (k, v["hbm_bytes_per_launch"] / 1e9))


if __name__ == "__main__":
    main()
