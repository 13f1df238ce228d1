#!/usr/bin/env python3
"""Encode time of the synthetic table with backslashes sprinkled into its strings (one letter in `--every` replaced
where the next byte is a letter too, so no field boundary changes): does the escape path of the tokeniser cost?
    python tools/escape_bench.py --gb 4 --every 100"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=4.0)
    ap.add_argument("--every", type=int, default=100)
    ap.add_argument("--before-tab", action="store_true", help="also put backslashes right before tabs' predecessors (\\\\ pairs: still no boundary change)")
    args = ap.parse_args()
    import torch
    from zdw_amd import capi, synth
    table = synth.DeviceTable(ncols=200, device="cuda:0")
    rows = table.rows_for_bytes(args.gb * 1e9)
    buf, nbytes = table.generate(0, rows)
    v = buf[:nbytes]
    for every in (0, args.every):
        if every:
            letter = (v >= 97) & (v <= 122)
            ok = letter[:-1] & letter[1:] & (torch.rand(nbytes - 1, device=v.device) < 1.0 / every)
            v[:-1][ok] = 0x5c
            del letter, ok
        enc = capi.Encoder(table.types, device=0)
        enc.set_input_device(buf.data_ptr(), nbytes, keepalive=buf)
        enc.enable_timing(True)
        for it in range(3):
            if it == 1:
                enc.enable_timing(True)
            rc, p1 = enc.pass1()
            assert rc == 0, rc
            enc.dict_section_device(); enc.stats(); enc.pass2_device()
        st = enc.stage_times()
        print("backslash every %d bytes:" % every if every else "no backslashes:", {k: round(x[0] / 2, 3) for k, x in st.items() if x[1]})
        enc.close()


if __name__ == "__main__":
    main()
