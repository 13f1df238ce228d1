import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from zdw_amd import capi, synth
table = synth.DeviceTable(ncols=1000, device="cuda:0", id_every=20)
rows = table.rows_for_bytes(2.0e9)
buf, n = table.generate(0, rows)
enc = capi.Encoder(table.types, device=0, mem_limit_mb=250)
enc.set_input_device(buf.data_ptr(), n, keepalive=buf)
prev=0
while True:
    rc, p1 = enc.pass1()
    l=enc.pass1_launches()
    print("block rows", p1.num_rows, "entries", p1.dict_entries, "last", p1.is_last_block, "launches", l-prev, "slots", p1.table_slots, "partial", enc.partial_fields(), flush=True)
    prev=l
    if p1.is_last_block: break
    enc.next_block()
