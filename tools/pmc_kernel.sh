#!/bin/bash
# counter passes for one kernel (name substring $1) of bench.py --gb 4 (separate runs; no trace options combined with --pmc)
k=$1; shift
out=gpurun_out/pmc_k
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify > $out/$name.json 2> $out/$name.err; echo "pass $name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_DATA_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum
run grbm GRBM_GUI_ACTIVE
python - "$k" <<'PY'
import csv, glob, collections, sys
k=sys.argv[1]
for f in sorted(glob.glob('gpurun_out/pmc_k/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float); n=0
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], {kk: '%.4g'%v for kk,v in agg.items()})
# one JSON for profiles/: counter -> sum over the kernel's dispatches (4 GB of the benchmark's table, one step)
import json
tot={}
for f in sorted(glob.glob('gpurun_out/pmc_k/*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            tot[r['Counter_Name']]=tot.get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
json.dump({"kernel": k, "workload": "bench.py --gb 4 --steps 1 (4.0 GB x 200 columns)", "counters": tot}, open('gpurun_out/pmc_k/summary.json','w'), indent=1, sort_keys=True)
PY
