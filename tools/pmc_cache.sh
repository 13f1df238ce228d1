#!/bin/bash
# cache counters of one kernel (name substring $1) of bench.py --gb 4: vector L1 (TCP) and L2 (TCC) requests, hits and misses
k=$1
out=gpurun_out/pmc_cache
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 5 120 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify > $out/$name.json 2> $out/$name.err; echo "pass $name rc=$?"; }
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcc2 TCC_READ_sum TCC_WRITE_sum TCC_EA0_WRREQ_sum
python - "$k" <<'PY'
import csv, glob, collections, sys
k=sys.argv[1]
for f in sorted(glob.glob('gpurun_out/pmc_cache/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if k in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], {kk: '%.4g'%v for kk,v in agg.items()})
PY
