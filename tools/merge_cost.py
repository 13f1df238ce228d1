#!/usr/bin/env python3
"""What does rank 0 of an N-rank run pay for the dictionary merge?  One GPU: N shards of G GB are pass-1'd
one after the other (each in its own context), their uniques exported; then shard 0 imports the other N-1
shards' strings and finishes its block.  Wall-clock per step of shard 0's part, with HIP-event stage times.

    python tools/merge_cost.py --shards 8 --gb 2
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--gb", type=float, default=2.0)
    ap.add_argument("--slots", type=int, default=1 << 26)
    ap.add_argument("--partition", action="store_true", help="time the range-partitioned dictionary instead of the all-gather merge")
    args = ap.parse_args()
    import torch
    from zdw_amd import capi, sharded, synth
    table = synth.DeviceTable(ncols=200, device="cuda:0")
    rows = table.rows_for_bytes(args.gb * 1e9)
    if args.partition:
        return partition_cost(args, table, rows)
    uniq = []
    enc0 = buf0 = None
    for r in range(args.shards):
        buf, nbytes = table.generate(r * rows, rows)
        torch.cuda.synchronize()
        enc = capi.Encoder(table.types, device=0, dict_slots=args.slots)
        enc.set_input_device(buf.data_ptr(), nbytes, keepalive=buf)
        rc, p1 = enc.pass1()
        assert rc == 0
        uniq.append(sharded.export_uniques_tensors(enc, "cuda:0"))
        if r == 0:
            enc0, buf0 = enc, buf
        else:
            enc.close()
            del buf
    print("uniques per shard:", [int(u[0].numel()) for u in uniq], "bytes:", [int(u[1].numel()) for u in uniq])
    lens, byts = sharded.concat_uniques(uniq, skip=0)
    torch.cuda.synchronize()
    for step in range(2):
        rc, p1 = enc0.pass1()
        enc0.enable_timing(True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sharded.export_uniques_tensors(enc0, "cuda:0")
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        enc0.import_uniques(lens.data_ptr(), byts.data_ptr(), lens.numel(), byts.numel())
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        _, dlen, _ = enc0.dict_section_device()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        enc0.stats()
        _, rlen = enc0.pass2_device()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print("step %d: export %.2f ms  import(%d strings) %.2f ms  dict %.2f ms (%.1f MB)  pass2 %.2f ms   stages %s"
              % (step, (t1 - t0) * 1e3, lens.numel(), (t2 - t1) * 1e3, (t3 - t2) * 1e3, dlen / 1e6, (t4 - t3) * 1e3,
                 {k: round(v[0], 2) for k, v in enc0.stage_times().items() if v[1]}))


def partition_cost(args, table, rows):
    """Rank 0's dictionary steps with the range-partitioned dictionary (the other shards only contribute the
    strings of key range 0 and their samples)."""
    import torch
    from zdw_amd import capi, sharded
    n = args.shards
    encs, bufs, samples = [], [], []
    for r in range(n):
        buf, nbytes = table.generate(r * rows, rows)
        torch.cuda.synchronize()
        enc = capi.Encoder(table.types, device=0)
        enc.set_input_device(buf.data_ptr(), nbytes, keepalive=buf)
        rc, p1 = enc.pass1()
        assert rc == 0
        enc.sort_uniques()
        samples += enc.sample_keys(sharded.SAMPLES_PER_SHARD)
        encs.append(enc)
        bufs.append(buf)
    splitters = sharded.pick_splitters(samples, n)
    to0_l, to0_b, mine_counts = [], [], None
    for r, enc in enumerate(encs):
        counts, nb, d_lens, d_bytes = enc.partition(splitters)
        lens = sharded._wrap(d_lens, sum(counts) * 4, "<i4", "cuda:0")
        byts = sharded._wrap(d_bytes, sum(nb), "|u1", "cuda:0")
        to0_l.append(lens[:counts[0]].clone())
        to0_b.append(byts[:nb[0]].clone())
        if r == 0:
            mine_counts = counts
        print("shard %d sends %s strings" % (r, counts))
    got_l, got_b = torch.cat(to0_l), torch.cat(to0_b)
    torch.cuda.synchronize()
    enc0 = encs[0]
    for step in range(2):
        rc, p1 = enc0.pass1()
        enc0.enable_timing(True)
        torch.cuda.synchronize()
        t = [time.perf_counter()]
        enc0.sort_uniques(); torch.cuda.synchronize(); t.append(time.perf_counter())
        enc0.sample_keys(sharded.SAMPLES_PER_SHARD); t.append(time.perf_counter())
        counts, nb, _, _ = enc0.partition(splitters); torch.cuda.synchronize(); t.append(time.perf_counter())
        sl = enc0.owner_build(got_l.data_ptr(), got_b.data_ptr(), got_l.numel(), got_b.numel()); torch.cuda.synchronize(); t.append(time.perf_counter())
        enc0.owner_offsets(1); torch.cuda.synchronize(); t.append(time.perf_counter())
        fake = torch.ones(sum(counts), dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        t.append(time.perf_counter())
        enc0.set_offsets(fake.data_ptr(), fake.numel(), 1 + sl * n, True); torch.cuda.synchronize(); t.append(time.perf_counter())
        names = ["sort_own", "sample", "partition+export", "owner_build(%d strings)" % got_l.numel(), "owner_offsets", "-", "set_offsets"]
        print("step %d: " % step + "  ".join("%s %.2f ms" % (nm, (t[i + 1] - t[i]) * 1e3) for i, nm in enumerate(names) if nm != "-"),
              " total %.2f ms; slice %.1f MB" % ((t[-1] - t[0]) * 1e3, sl / 1e6))


if __name__ == "__main__":
    main()
