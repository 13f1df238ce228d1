#!/bin/bash
# instruction-cache and issue counters of pass 1 (bench.py --gb 4 --pass1-only): does the kernel's code fit the 64 KiB two CUs share?
out=gpurun_out/pmc_ic
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python bench.py --gb 4 --steps 1 --warmup 0 --pass1-only --no-cpu --no-e2e --no-verify > $out/$name.json 2> $out/$name.err; echo "pass $name rc=$?"; }
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
run dc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_ic/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if ('k_pass1<' in r['Kernel_Name'] or 'k_pass1_rows' in r['Kernel_Name']): agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print({k:'%.4g'%v for k,v in agg.items()})
PY
