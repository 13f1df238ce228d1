#!/usr/bin/env python3
"""Random tables through the C ABI against the CPU oracle for a given time (test infrastructure: run on the GPU box).

    python tools/fuzz_oracle.py --seconds 240 [--seed 1]

Every case draws a schema (1 .. 3000 columns of the types the reference knows), a shape (row count, share of empty fields, repeats)
and a set of nuisances: escaped tabs / newlines / backslashes, fields of up to 200 KB, blank lines, numbers with signs, blanks and
junk, values beyond their type, char(1) with escapes, bytes >= 0x80, an unterminated last row; sometimes -t, a forced block split or
the kernel form for mostly empty fields.  A mismatch is written to gpurun_out/fuzz/ with everything needed to replay it."""
import argparse
import os
import pickle
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle          # noqa: E402  (the checker)
from tests import parity_util as pu, zdw_walk          # noqa: E402
from zdw_amd import capi          # noqa: E402

SQL = [("varchar(255)", 0), ("int(11) unsigned", 9), ("bigint(20) unsigned", 10), ("smallint(5) unsigned", 8), ("tinyint(3) unsigned", 7),
       ("int(11)", 14), ("char(1)", 6), ("datetime", 2), ("decimal(24,12)", 11)]
ALPHA = b"abcdefghijklmnopqrstuvwxyz0123456789_ -.:/"


def field(rnd, sql, nuisance):
    r = rnd.random()
    if sql.startswith("varchar") or sql in ("datetime", "decimal(24,12)"):
        if r < nuisance * 0.02:
            n = rnd.choice([25, 30, 64, 255, 256, 300, 1000, 5000, 70000, 200000])
            return bytes(rnd.choice(ALPHA) for _ in range(min(n, 64))) * (n // 64 + 1)
        if r < nuisance * 0.10:
            return rnd.choice([b"\\\t", b"\\\n", b"\\\\", b"a\\\tb", b"\\\\\\\t", b"x\\\\", b"\\", b"end\\\\"]) + bytes(rnd.choice(ALPHA) for _ in range(rnd.randrange(0, 9)))
        if r < nuisance * 0.14:
            return bytes(rnd.choice(b"\xc3\xa9\xff\x80z") for _ in range(rnd.randrange(1, 30)))
        if sql == "datetime":
            return b"2024-%02d-%02d %02d:%02d:%02d" % (rnd.randrange(1, 13), rnd.randrange(1, 29), rnd.randrange(24), rnd.randrange(60), rnd.randrange(60))
        if sql.startswith("decimal"):
            return b"%d.%012d" % (rnd.randrange(100000), rnd.randrange(10 ** 12))
        n = rnd.choice([1, 2, 3, 5, 8, 9, 15, 16, 17, 23, 24, 25, 31, 40])
        return bytes(rnd.choice(ALPHA[:38]) for _ in range(n)) if rnd.random() < 0.3 else b"w%d" % rnd.randrange(rnd.choice([5, 50, 5000]))
    if sql == "char(1)":
        return rnd.choice([b"a", b"b", b"z", b"\\\t", b"\\\\", b"\\\n", b"ab", b"\xff", b"0", b" "]) if r < 0.5 else bytes([rnd.randrange(33, 127)]).replace(b"\\", b"\\\\")
    # integers
    if r < nuisance * 0.08:
        return rnd.choice([b"-1", b"+7", b" 12", b"12 ", b"1e5", b"0x10", b"99999999999999999999999", b"18446744073709551615", b"18446744073709551616",
                           b"4294967296", b"00000000000000000123", b"12345678901234567", b"1234567890123456", b"-", b"abc", b"9" * 30, b"\\\t5"])
    bits = rnd.choice([3, 7, 15, 31, 47, 63])
    return b"%d" % rnd.getrandbits(bits)


def make_case(rnd):
    ncols = rnd.choice([1, 2, 3, 5, 8, 17, 33, 64, 65, 130, 200, 400, 900, 3000]) if rnd.random() < 0.8 else rnd.randrange(1, 700)
    budget = rnd.choice([2000, 20000, 200000, 1500000])          # fields
    nrows = max(1, min(budget // ncols, rnd.choice([1, 3, 50, 700, 5000, 40000])))
    p_empty = rnd.choice([0.0, 0.1, 0.25, 0.5, 0.9, 0.98])
    p_repeat = rnd.choice([0.0, 0.3, 0.8])
    nuisance = rnd.choice([0.0, 0.0, 0.3, 1.0])
    kinds = [rnd.choice(SQL) for _ in range(ncols)] if rnd.random() < 0.7 else [rnd.choice(SQL[:2])] * ncols
    desc = b"".join(b"c%d\t%s\n" % (i, k[0].encode()) for i, k in enumerate(kinds))
    types = [k[1] for k in kinds]
    rows, prev = [], [b""] * ncols
    for r in range(nrows):
        cur = []
        for c in range(ncols):
            if r and rnd.random() < p_repeat:
                cur.append(prev[c])
            elif rnd.random() < p_empty:
                cur.append(b"")
            else:
                cur.append(field(rnd, kinds[c][0], nuisance))
        rows.append(b"\t".join(cur) + b"\n")
        if nuisance and rnd.random() < 0.01:
            rows.append(b"\n")                                   # a blank line is no row
        prev = cur
    tsv = b"".join(rows)
    if nuisance and rnd.random() < 0.2 and tsv.endswith(b"\n"):
        tsv = tsv[:-1]                                           # unterminated last row
    trim = rnd.random() < 0.15
    split = []
    if nrows > 4 and rnd.random() < 0.2:
        split = [rnd.randrange(1, nrows)]
    env = {}
    if rnd.random() < 0.25:
        env["ZDW_P1_SKIP_EMPTY"] = rnd.choice(["0", "1"])
    if rnd.random() < 0.1:
        env["ZDW_P1_ALL_IN_ONE"] = "1"
    if rnd.random() < 0.15:
        env["ZDW_P1_WPB"] = rnd.choice(["4", "12", "16"])
    if rnd.random() < 0.3:
        env["ZDW_P1_ROWS"] = "2"                                 # pass 1's second form (a lane per row) wherever it can run
        if rnd.random() < 0.5:
            env["ZDW_P1_ROWS_PER_LANE"] = rnd.choice(["1", "2", "7"])
    shards = rnd.choice([2, 3, 5]) if rnd.random() < 0.3 else 0
    if rnd.random() < 0.35:
        env["ZDW_SORT_PACK"] = "1"                               # the sort's packed first round wherever 8 characters fit one key
    if rnd.random() < 0.35:
        env["ZDW_SORT_SCATTER"] = "1"                            # the radix passes' second form (chunks put in order in LDS)
    return dict(desc=desc, types=types, tsv=tsv, trim=trim, split=split, env=env, ncols=ncols, nrows=nrows, shards=shards)


def run_case(c):
    rc0, want, info = oracle.encode(c["desc"], c["tsv"], trim=c["trim"], split_rows=c["split"])
    saved = {k: os.environ.get(k) for k in c["env"]}
    os.environ.update(c["env"])
    try:
        enc = capi.Encoder(c["types"], trim=c["trim"])
        enc.set_input_host(c["tsv"])
        got_blocks, rc = [], 0
        limits = list(c["split"])
        while True:
            try:
                rc, blk, p1 = enc.encode_block(max_rows=limits.pop(0) if limits else None)
            except capi.ZdwError as e:
                rc, blk = e.code or 1, None
            if rc or blk is None:
                break
            got_blocks.append(blk.tobytes())
            if p1.is_last_block:
                break
            enc.next_block()
        enc.close()
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    if rc0 != 0:
        return (rc != 0), "oracle rc %d, library rc %d" % (rc0, rc)
    if rc != 0:
        return False, "library rc %d where the oracle encodes" % rc
    z = zdw_walk.walk(want)
    got = want[:z.header_len] + b"".join(got_blocks)
    if got != want:
        return False, pu.first_difference(got, want)
    if c.get("shards") and not c["split"]:
        # the same block row-sharded over several contexts inside the library (the same device listed several times)
        m = capi.MultiEncoder(c["types"], [0] * c["shards"], trim=c["trim"])
        try:
            rc, body, blocks = m.encode_all(c["tsv"])
        except capi.ZdwError as e:
            rc, body = e.code or 1, None
        m.close()
        if rc != 0:
            return False, "%d shards: rc %d" % (c["shards"], rc)
        got = want[:z.header_len] + body
        if got != want:
            return False, "%d shards: %s" % (c["shards"], pu.first_difference(got, want))
    return True, ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240.0)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    os.makedirs("gpurun_out/fuzz", exist_ok=True)
    rnd = random.Random(args.seed)
    t0, n, bad, nbytes = time.time(), 0, 0, 0
    while time.time() - t0 < args.seconds:
        c = make_case(rnd)
        ok, why = run_case(c)
        n += 1
        nbytes += len(c["tsv"])
        if not ok:
            bad += 1
            path = "gpurun_out/fuzz/fail_%d_%d.pkl" % (args.seed, n)
            pickle.dump(c, open(path, "wb"))
            print("MISMATCH case %d (%d cols x %d rows, trim %s, split %s, env %s): %s -> %s" % (n, c["ncols"], c["nrows"], c["trim"], c["split"], c["env"], why, path), flush=True)
            if bad >= 5:
                break
        if n % 50 == 0:
            print("%d cases, %.1f MB, %d mismatches, %.0f s" % (n, nbytes / 1e6, bad, time.time() - t0), flush=True)
    print("done: %d cases, %.1f MB, %d mismatches" % (n, nbytes / 1e6, bad), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
