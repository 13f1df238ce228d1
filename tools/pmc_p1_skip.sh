#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_s; rm -rf $out; mkdir -p $out
for sk in 32 1 0; do
  export ZDW_DEBUG_SKIP=$sk
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $out/s$sk -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu --pass1-only > $out/s$sk.json 2> $out/s$sk.err
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_s/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pass1' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], {k: '%.4g'%v for k,v in sorted(agg.items())})
PY
