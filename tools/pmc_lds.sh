#!/bin/bash
# LDS counters of pass 1 (bench.py --gb 4 --pass1-only): how busy is the LDS unit of a CU?
out=gpurun_out/pmc_lds
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python bench.py --gb 4 --steps 1 --warmup 0 --pass1-only --no-cpu --no-e2e --no-verify > $out/$name.json 2> $out/$name.err; echo "pass $name rc=$?"; }
run a SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run b SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_lds/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pass1<' in r['Kernel_Name']: agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print({k:'%.4g'%v for k,v in agg.items()})
PY
