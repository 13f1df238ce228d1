#!/bin/bash
# pass 1 alone (bench.py --pass1-only, 4 GB) under rocprofv3 --pmc, once per ZDW_DEBUG_SKIP setting ($@; default: 1 0):
# 1 = tokeniser + descriptor handling without field work, 2 = no dictionary probes, 0 = everything
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_s; rm -rf $out; mkdir -p $out
for sk in ${@:-1 0}; do
  export ZDW_DEBUG_SKIP=$sk
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES --output-format csv -d $out/s$sk -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify --pass1-only $BENCH_ARGS > $out/s$sk.json 2> $out/s$sk.err
  timeout -k 5 120 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/t$sk -- python bench.py --gb 4 --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify --pass1-only $BENCH_ARGS > $out/t$sk.json 2> $out/t$sk.err
done
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/pmc_s/*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'k_pass1<' in r['Kernel_Name']:
            agg[r['Counter_Name']]+=float(r['Counter_Value'])
    print(f.split('/')[2], {k: '%.4g'%v for k,v in sorted(agg.items())})
PY
