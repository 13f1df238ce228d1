#!/usr/bin/env python3
"""Two halves of tools/run_decode.py, so that the decoding half can run under rocprofv3 (which must not see a child process start):
    python tools/dec_file.py make /tmp/t.zdw --gb 4        the benchmark's table through the shipped convertDWfile (compressor = cat)
    python tools/dec_file.py decode /tmp/t.zdw             that raw stream through zdw_dec_* twice; stage times of the second run"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["make", "decode"])
    ap.add_argument("path")
    ap.add_argument("--gb", type=float, default=4.0)
    ap.add_argument("--cols", type=int, default=200)
    args = ap.parse_args()
    if args.mode == "make":
        import torch
        from zdw_amd import build, synth
        table = synth.DeviceTable(ncols=args.cols, device="cuda:0")
        rows = table.rows_for_bytes(args.gb * 1e9)
        buf, n = table.generate(0, rows)
        d = tempfile.mkdtemp(prefix="zdwdec_")
        try:
            with open(os.path.join(d, "t.sql"), "wb") as f:
                for o in range(0, n, 1 << 28):
                    f.write(buf[o:min(n, o + (1 << 28))].cpu().numpy().tobytes())
            open(os.path.join(d, "t.desc.sql"), "wb").write(table.desc)
            shim = os.path.join(d, "shim")
            os.makedirs(shim)
            with open(os.path.join(shim, "gzip"), "w") as f:
                f.write("#!/bin/sh\nexec cat\n")
            os.chmod(os.path.join(shim, "gzip"), 0o755)
            env = dict(os.environ, ZDW_GZIP_THREADS="0", PATH=shim + os.pathsep + os.environ["PATH"])
            del buf
            torch.cuda.empty_cache()
            r = subprocess.run([build.CLI_EXE, "-q", os.path.join(d, "t.sql")], env=env)
            assert r.returncode == 0
            shutil.move(os.path.join(d, "t.zdw.gz"), args.path)
            print("made", args.path, os.path.getsize(args.path), "bytes;", rows, "rows")
        finally:
            shutil.rmtree(d, ignore_errors=True)
        return
    from zdw_amd import capi
    z = open(args.path, "rb").read()
    dec = capi.Decoder(0)
    for i in range(2):
        assert dec.set_input(z) == 0, dec.error()
        stages, text = {}, 0
        while True:
            rc, blk = dec.next_block()
            assert rc == 0, dec.error()
            if blk.num_rows:
                text += blk.text_bytes
                for k, v in dec.stage_ms().items():
                    stages[k] = round(stages.get(k, 0.0) + v, 3)
            if blk.is_last_block:
                break
    print({"text_bytes": int(text), "stage_ms": stages})


if __name__ == "__main__":
    main()
