"""tools/rows_ab.py [GB] -- the two forms of pass 1 side by side: small tables against the checker with the lane-per-row form forced,
then a table of the benchmark's shape in HBM encoded by both forms (same bytes? stage times)."""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def small():
    from oracle import oracle
    from zdw_amd import capi, synth
    bad = 0
    for seed, ncols, nrows in [(1, 20, 300), (2, 200, 700), (3, 33, 5000), (4, 7, 900), (5, 200, 3000), (6, 64, 20000)]:
        desc, types, tsv = synth.make_table(seed=seed, ncols=ncols, nrows=nrows)
        rc, want, info = oracle.encode(desc, tsv)
        for form in ("0", "2"):
            os.environ["ZDW_P1_ROWS"] = form
            enc = capi.Encoder(types)
            enc.set_input_host(tsv)
            rc2, blk, _ = enc.encode_block()
            got = blk.tobytes() if blk is not None else b"rc%d" % rc2
            enc.close()
            ok = got == want[info.header_bytes:]
            print("small seed %d %3d cols %6d rows form %s: %s" % (seed, ncols, nrows, form, "ok" if ok else "MISMATCH"), flush=True)
            if not ok:
                bad += 1
                n = min(len(got), len(want) - info.header_bytes)
                i = next((k for k in range(n) if got[k] != want[info.header_bytes + k]), n)
                print("   first difference at %d of %d / %d" % (i, len(got), len(want) - info.header_bytes))
    return bad


def big(gb):
    import torch
    from zdw_amd import capi, synth
    table = synth.DeviceTable(ncols=200, device="cuda:0")
    rows = table.rows_for_bytes(gb * 1e9)
    buf, nbytes = table.generate(0, rows)
    torch.cuda.synchronize()
    out = {}
    for form in ("0", "1"):
        os.environ["ZDW_P1_ROWS"] = form
        enc = capi.Encoder(table.types, device=0)
        enc.set_input_device(buf.data_ptr(), nbytes, keepalive=buf)
        enc.enable_timing(True)
        best = None
        for it in range(4):
            if it == 1:
                enc.enable_timing(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc, p1 = enc.pass1()
            assert rc == 0, rc
            dptr, dlen, _ = enc.dict_section_device()
            stats = enc.stats()
            rptr, rlen = enc.pass2_device()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None or dt < best else best
        st = {k: round(v[0] / 3, 3) for k, v in enc.stage_times().items() if v[1]}
        dsec, _ = enc.dict_section()
        rws = enc.pass2()
        h = hashlib.sha256(bytes(dsec)).hexdigest()[:16], hashlib.sha256(bytes(stats[0]) + repr(list(stats[1])).encode()).hexdigest()[:16], hashlib.sha256(bytes(rws)).hexdigest()[:16]
        out[form] = h
        print("form %s: %.2f GB, %d rows, step %.2f ms, stages %s, sha dict/stats/rows %s" % (form, nbytes / 1e9, p1.num_rows, best * 1e3, st, h), flush=True)
        enc.close()
    print("same bytes:", out["0"] == out["1"])
    return 0 if out["0"] == out["1"] else 1


if __name__ == "__main__":
    gb = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
    bad = small()
    if gb > 0:
        bad += big(gb)
    print("FAILED" if bad else "all good")
    sys.exit(1 if bad else 0)
