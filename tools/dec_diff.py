#!/usr/bin/env python3
"""GPU decode of a golden file against its committed source: where the first difference is (debugging aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity_util as pu
from zdw_amd import capi
name = sys.argv[1] if len(sys.argv) > 1 else "movie_tickets"
d = capi.Decoder(0)
rc, text, blocks = d.decode_all(pu.rd(name + ".zdw.xz"))
want = pu.rd(name + ".sql.xz")
print("rc", rc, len(text), len(want))
n = min(len(text), len(want))
i = next((k for k in range(n) if text[k] != want[k]), n)
print("first difference at", i)
print("got ", text[max(0, i - 80):i + 80])
print("want", want[max(0, i - 80):i + 80])
ndiff = sum(1 for k in range(n) if text[k] != want[k])
print("differing bytes", ndiff)
