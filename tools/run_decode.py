#!/usr/bin/env python3
"""GPU decode (zdw_dec_*) of the benchmark's table: --gb of synthetic TSV through the shipped convertDWfile (compressor = cat), the raw
.zdw stream back through the decoder on the device, twice (the second run has its buffers); device time by stage, GB/s of text.

    python tools/run_decode.py --gb 10 --out gpurun_out/decode.json [--check]      (--check: sha256 against zdw_amd/lib/unconvertDWfile)"""
import argparse
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=10.0)
    ap.add_argument("--cols", type=int, default=200)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--out", default="gpurun_out/decode.json")
    args = ap.parse_args()
    import torch
    from zdw_amd import build, capi, synth
    table = synth.DeviceTable(ncols=args.cols, device="cuda:0")
    rows = table.rows_for_bytes(args.gb * 1e9)
    buf, n = table.generate(0, rows)
    d = tempfile.mkdtemp(prefix="zdwdec_")
    try:
        with open(os.path.join(d, "t.sql"), "wb") as f:
            for o in range(0, n, 1 << 28):
                f.write(buf[o:min(n, o + (1 << 28))].cpu().numpy().tobytes())
        open(os.path.join(d, "t.desc.sql"), "wb").write(table.desc)
        del buf
        torch.cuda.empty_cache()
        shim = os.path.join(d, "shim")
        os.makedirs(shim)
        with open(os.path.join(shim, "gzip"), "w") as f:
            f.write("#!/bin/sh\nexec cat\n")
        os.chmod(os.path.join(shim, "gzip"), 0o755)
        env = dict(os.environ, ZDW_GZIP_THREADS="0", PATH=shim + os.pathsep + os.environ["PATH"])
        r = subprocess.run([build.CLI_EXE, "-q", os.path.join(d, "t.sql")], env=env)
        assert r.returncode == 0
        os.unlink(os.path.join(d, "t.sql"))
        zpath = os.path.join(d, "t.zdw")
        os.rename(os.path.join(d, "t.zdw.gz"), zpath)          # (the compressor was cat: a raw stream)
        z = open(zpath, "rb").read()
        dec = capi.Decoder(0)
        runs = []
        for i in range(3):
            t0 = time.perf_counter()
            rc = dec.set_input(z)
            assert rc == 0, dec.error()
            t1 = time.perf_counter()
            blocks, stages, text_bytes = 0, {}, 0
            sha = hashlib.sha256() if (args.check and i == 0) else None
            while True:
                rc, blk = dec.next_block()
                assert rc == 0, dec.error()
                if blk.num_rows:
                    blocks += 1
                    text_bytes += blk.text_bytes
                    for k, v in dec.stage_ms().items():
                        stages[k] = round(stages.get(k, 0.0) + v, 3)
                    passes = blk.walk_passes
                    if sha is not None:
                        sha.update(dec.text())
                if blk.is_last_block:
                    break
            t2 = time.perf_counter()
            dev_ms = sum(stages.values())
            runs.append({"upload_ms": round((t1 - t0) * 1e3, 1), "decode_wall_ms": round((t2 - t1) * 1e3, 1), "device_ms": round(dev_ms, 2), "stage_ms": stages,
                         "text_GBps_device": round(text_bytes / 1e6 / dev_ms, 1) if dev_ms else None, "walk_passes": int(passes), "blocks": blocks})
            print(runs[-1], flush=True)
            if sha is not None:
                got = sha.hexdigest()
        res = {"workload": "decode of the synthetic table %.1f GB x %d cols (BASELINE configs[3] shape): %d bytes of ZDW -> %d bytes of TSV" % (n / 1e9, args.cols, len(z), text_bytes),
               "zdw_bytes": len(z), "text_bytes": int(text_bytes), "rows": int(rows), "runs": runs}
        if args.check:
            t0 = time.perf_counter()
            p = subprocess.run("%s - %s | sha256sum" % (build.DECODER_EXE, zpath), shell=True, capture_output=True)
            res["host_decoder"] = {"seconds": round(time.perf_counter() - t0, 1), "same_sha256": p.stdout.split()[0].decode() == got}
            print(res["host_decoder"])
            assert res["host_decoder"]["same_sha256"]
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
