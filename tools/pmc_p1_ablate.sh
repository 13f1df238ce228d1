#!/bin/bash
# Instruction counts of pass 1 by part (bench.py --gb 4): the experiments build (tools/build_experiments.sh) with work removed --
# ZDW_DEBUG_SKIP = 32 the tokeniser alone, 1 descriptors handled but no field work, 2 no dictionary probes, 0 everything.
out=gpurun_out/pmc_ablate
rm -rf $out; mkdir -p $out
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for skip in 0 32 1 2; do
  export ZDW_DEBUG_SKIP=$skip
  timeout -k 5 120 python bench.py --gb 4 --steps 3 --warmup 1 --pass1-only --no-cpu --no-e2e --no-verify > $out/time_$skip.json 2> $out/time_$skip.err
  echo "skip $skip: $(grep -o '"pass1": [0-9.]*' $out/time_$skip.json)"
  timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $out/s$skip -- python bench.py --gb 4 --steps 1 --warmup 0 --pass1-only --no-cpu --no-e2e --no-verify > $out/pmc_$skip.json 2> $out/pmc_$skip.err
  echo "  pmc rc=$?"
done
python - <<'PY'
import csv, glob, collections, json
res={}
for skip in (0,32,1,2):
    agg=collections.defaultdict(float)
    for f in glob.glob('gpurun_out/pmc_ablate/s%d/**/*counter_collection.csv'%skip, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_pass1<' in r['Kernel_Name']:
                agg[r['Counter_Name']]+=float(r['Counter_Value'])
    res[str(skip)]=dict(agg)
    print(skip, {k:'%.4g'%v for k,v in agg.items()})
json.dump(res, open('gpurun_out/pmc_ablate/summary.json','w'), indent=1, sort_keys=True)
PY
