#!/usr/bin/env python3
"""Runs the reference-generated vectors through zdw_amd/lib/convertDWfile as CHILD processes and prints the
first failures with their stderr (debugging aid for crashes that take the in-process test runner down)."""
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity_util as pu
from zdw_amd import build

bad = 0
for fname in ("edge_cases.json", "fuzz_cases.json"):
    for c in pu.cases(fname):
        d = tempfile.mkdtemp(prefix="zc_")
        shim = os.path.join(d, "shim"); os.makedirs(shim)
        open(os.path.join(shim, "gzip"), "w").write("#!/bin/sh\nexec cat\n"); os.chmod(os.path.join(shim, "gzip"), 0o755)
        stem = os.path.join(d, "t")
        open(stem + ".sql", "wb").write(c["tsv"]); open(stem + ".desc.sql", "wb").write(c["desc"])
        if c["meta_file"] is not None:
            open(stem + ".metadata", "wb").write(c["meta_file"])
        env = dict(os.environ); env["PATH"] = shim + os.pathsep + env["PATH"]; env["ZDW_GZIP_THREADS"] = "0"
        p = subprocess.run([build.CLI_EXE] + c["args"] + [stem + ".sql"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        out = open(stem + ".zdw.gz", "rb").read() if os.path.exists(stem + ".zdw.gz") else None
        ok = p.returncode == c["exit_code"] and (c["exit_code"] != 0 or out == c["zdw"])
        if not ok:
            bad += 1
            print("FAIL", c["name"], "rc", p.returncode, "want", c["exit_code"], "| stderr:", p.stderr[-600:].decode(errors="replace"))
            if bad >= 5:
                sys.exit(1)
        shutil.rmtree(d, ignore_errors=True)
print("all cases ok" if not bad else "%d failures" % bad)
