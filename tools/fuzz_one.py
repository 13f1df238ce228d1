#!/usr/bin/env python3
"""tools/fuzz_one.py SEED N [N ...]: the N-th case(s) of tools/fuzz_oracle.py's sequence for SEED, alone (the cases before them are drawn, not run)."""
import importlib.util
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("fuzz_oracle", os.path.join(ROOT, "tools", "fuzz_oracle.py"))
fz = importlib.util.module_from_spec(spec)
spec.loader.exec_module(fz)

seed, want = int(sys.argv[1]), sorted(int(x) for x in sys.argv[2:])
rnd = random.Random(seed)
bad = 0
for n in range(1, want[-1] + 1):
    c = fz.make_case(rnd)
    if n in want:
        print("case %d: %d cols x %d rows, %d bytes, split %s, env %s" % (n, c["ncols"], c["nrows"], len(c["tsv"]), c["split"], c["env"]), flush=True)
        ok, why = fz.run_case(c)
        print("   ", "ok" if ok else "MISMATCH: " + why[:300], flush=True)
        bad += not ok
sys.exit(1 if bad else 0)
