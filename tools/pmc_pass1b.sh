#!/bin/bash
out=gpurun_out/pmc_p1b
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; echo "pass $name"; timeout -k 5 100 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python bench.py --gb 2 --steps 1 --warmup 0 --no-cpu > $out/$name.json 2> $out/$name.err; echo "rc=$?"; }
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum
run tcp TCP_TOTAL_ACCESSES_sum TCP_TOTAL_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_p1b/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pass1' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], dict(agg))
PY
