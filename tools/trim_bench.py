#!/usr/bin/env python3
"""Stage times of the synthetic table with and without -t (trailing-space trimming: the all-in-one pass-1 kernel).
    python tools/trim_bench.py --gb 4"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=4.0)
    args = ap.parse_args()
    from zdw_amd import capi, synth
    table = synth.DeviceTable(ncols=200, device="cuda:0")
    rows = table.rows_for_bytes(args.gb * 1e9)
    buf, nbytes = table.generate(0, rows)
    for trim in (False, True):
        enc = capi.Encoder(table.types, device=0, trim=trim)
        enc.set_input_device(buf.data_ptr(), nbytes, keepalive=buf)
        enc.enable_timing(True)
        for it in range(3):
            if it == 1:
                enc.enable_timing(True)
            rc, p1 = enc.pass1()
            assert rc == 0, rc
            enc.dict_section_device(); enc.stats(); enc.pass2_device()
        st = enc.stage_times()
        print("trim" if trim else "plain", {k: round(x[0] / 2, 3) for k, x in st.items() if x[1]})
        enc.close()


if __name__ == "__main__":
    main()
