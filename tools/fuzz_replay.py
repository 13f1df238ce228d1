#!/usr/bin/env python3
"""tools/fuzz_replay.py [files]: the cases tools/fuzz_oracle.py saved as mismatches (gpurun_out/fuzz/fail_*.pkl), run again."""
import glob
import importlib.util
import os
import pickle
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fuzz_oracle", os.path.join(ROOT, "tools", "fuzz_oracle.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)

bad = 0
for f in sorted(sys.argv[1:] or glob.glob(os.path.join(ROOT, "gpurun_out", "fuzz", "fail_*.pkl"))):
    c = pickle.load(open(f, "rb"))
    ok, why = fz.run_case(c)
    print(os.path.basename(f), "ok" if ok else "MISMATCH: " + why[:200], c["env"], flush=True)
    bad += not ok
sys.exit(1 if bad else 0)
