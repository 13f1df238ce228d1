"""Where the one-block-over-N-ranks driver (zdw_amd/sharded.py: encode_shard_dist) spends its step, measured with ONE rank over RCCL on the
benchmark's 10 GB table: every phase between device synchronisations (python tools/dist_prof.py).  The phases that do not shrink with the
number of ranks -- the owner's build of the strings that arrive, the collectives -- are what weak scaling loses against the plain path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544"); os.environ.setdefault("RANK","0"); os.environ.setdefault("WORLD_SIZE","1")
import torch, torch.distributed as dist
from zdw_amd import capi, sharded, synth
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda",0))
table = synth.DeviceTable(ncols=200, device="cuda:0")
rows = table.rows_for_bytes(10e9)
buf, n = table.generate(0, rows)
enc = capi.Encoder(table.types, device=0)
enc.set_input_device(buf.data_ptr(), n, keepalive=buf)
import types
# wrap functions with timers
T = {}
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t=time.perf_counter(); r=f(*a, **k); torch.cuda.synchronize(); T[name]=T.get(name,0)+(time.perf_counter()-t)*1e3; return r
    setattr(mod, name, g)
for nm in ("allreduce_ranges","partitioned_dictionary","exchange_prev_row","allgather_i64","alltoallv","exchange_counts","pick_splitters"):
    wrap(sharded, nm)
for nm in ("pass1","sort_uniques","sample_keys","partition","owner_build","owner_offsets","set_offsets","last_row","line_caps","dict_section_device","stats","pass2_device","export_ranges","import_ranges","set_prev_row"):
    f = getattr(capi.Encoder, nm)
    def mk(f, nm):
        def g(self,*a,**k):
            torch.cuda.synchronize(); t=time.perf_counter(); r=f(self,*a,**k); torch.cuda.synchronize(); T["enc."+nm]=T.get("enc."+nm,0)+(time.perf_counter()-t)*1e3; return r
        return g
    setattr(capi.Encoder, nm, mk(f, nm))
for i in range(3):
    T.clear()
    torch.cuda.synchronize(); t0=time.perf_counter()
    sharded.encode_shard_dist(enc, 0, 1)
    torch.cuda.synchronize(); tot=(time.perf_counter()-t0)*1e3
print("total %.2f ms" % tot)
for k,v in sorted(T.items(), key=lambda kv:-kv[1]): print("  %-28s %.2f" % (k, v))
dist.destroy_process_group()
