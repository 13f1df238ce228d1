#!/bin/bash
# counter passes for k_pass1 (separate runs; no trace options combined with --pmc)
out=gpurun_out/pmc_p1
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu > $out/$name.json 2> $out/$name.err; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT
run ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES TA_FLAT_WAVEFRONTS
run tcp TCP_TOTAL_ACCESSES TCP_TOTAL_READ TCP_PENDING_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_p1/*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if 'k_pass1' in r['Kernel_Name']:
                agg[r['Counter_Name']]+=float(r['Counter_Value'])
        print(d.split('/')[-2], dict(agg))
PY
