#!/usr/bin/env python3
"""End-to-end time of the convertDWfile binary on a synthetic table written to local disk (page-cache warm),
with the compressor child replaced by `cat` and with real gzip.   python tools/cli_e2e.py --gb 1"""
import argparse, os, shutil, stat, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=1.0)
    ap.add_argument("--no-gzip", action="store_true")
    ap.add_argument("--cols", type=int, default=200)
    ap.add_argument("--id-every", type=int, default=0, help="every k-th column an id column (C5: --cols 1000 --id-every 20)")
    ap.add_argument("--mem-limit", type=int, default=0, help="--mem-limit=<MB> for the runs (0: the tool's default)")
    args = ap.parse_args()
    import torch
    from zdw_amd import build, synth
    table = synth.DeviceTable(ncols=args.cols, device="cuda:0", id_every=args.id_every)
    rows = table.rows_for_bytes(args.gb * 1e9)
    buf, nbytes = table.generate(0, rows)
    d = tempfile.mkdtemp(prefix="zdwcli_")
    try:
        with open(os.path.join(d, "t.sql"), "wb") as f:
            for o in range(0, nbytes, 1 << 28):
                f.write(buf[o:min(nbytes, o + (1 << 28))].cpu().numpy().tobytes())
        with open(os.path.join(d, "t.desc.sql"), "wb") as f:
            f.write(table.desc)
        del buf
        torch.cuda.empty_cache()
        shim = os.path.join(d, "shim"); os.makedirs(shim)
        p = os.path.join(shim, "gzip")
        with open(p, "w") as f:
            f.write("#!/bin/sh\nexec cat\n")
        os.chmod(p, os.stat(p).st_mode | stat.S_IXUSR)
        runs = [("cat", shim + os.pathsep + os.environ["PATH"], {"ZDW_GZIP_THREADS": "0"}),
                ("cat, file streamed in pieces", shim + os.pathsep + os.environ["PATH"], {"ZDW_CLI_STREAM_FILE": "1", "ZDW_GZIP_THREADS": "0"}),
                ("cat", shim + os.pathsep + os.environ["PATH"], {"ZDW_GZIP_THREADS": "0"})]
        if not args.no_gzip:
            runs.append(("gzip in the process", os.environ["PATH"], {}))
            runs.append(("gzip child", os.environ["PATH"], {"ZDW_GZIP_THREADS": "0"}))
        for label, path, extra in runs:
            env = dict(os.environ, PATH=path, ZDW_CLI_TIMING="1", **extra)
            t0 = time.perf_counter()
            r = subprocess.run([build.CLI_EXE, "-q"] + (["--mem-limit=%d" % args.mem_limit] if args.mem_limit else []) + [os.path.join(d, "t.sql")], env=env)
            dt = time.perf_counter() - t0
            out = os.path.getsize(os.path.join(d, "t.zdw.gz"))
            print("%s: rc=%d  %.2f s  %.1f MB/s of input  (%d bytes in, %d out)" % (label, r.returncode, dt, nbytes / 1e6 / dt, nbytes, out))
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
