#!/bin/bash
# A/B of k_pass1 counters: current tree vs the worktree in _prev
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/pmc_ab
i=0
for tree in . _prev; do
  for grp in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA_RDREQ_sum" "TCC_EA_WRREQ_sum TCC_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_NC_READ_REQ_sum TCP_TCC_UC_READ_REQ_sum"; do
    i=$((i+1))
    out=$GRAFT_REPO_ROOT/gpurun_out/pmc_ab/$(echo $tree | tr -d './_')x_$i
    mkdir -p $out
    (cd $tree && timeout -k 5 120 rocprofv3 --pmc $grp --output-format csv -d $out -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu > $out/b.json 2> $out/b.err)
  done
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_ab/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pass1' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], {k: '%.4g'%v for k,v in sorted(agg.items())})
PY
