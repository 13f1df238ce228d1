#!/bin/bash
# SQ counters of the decoder's kernels (4 GB of the benchmark's table): gpurun_out/pmc_dec/summary.json
out=gpurun_out/pmc_dec
rm -rf $out; mkdir -p $out
python tools/dec_file.py make /tmp/zdw_dec_prof.zdw --gb 4 > $out/make.log 2>&1 || { tail -n 5 $out/make.log; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 5 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- python tools/dec_file.py decode /tmp/zdw_dec_prof.zdw > $out/$name.json 2> $out/$name.err; echo "pass $name rc=$?"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS
run sq2 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SMEM
run grbm GRBM_GUI_ACTIVE
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python tools/dec_file.py decode /tmp/zdw_dec_prof.zdw > $out/trace.json 2> $out/trace.err
rm -f /tmp/zdw_dec_prof.zdw
python - <<'PY'
import csv, glob, json
tot={}
for f in sorted(glob.glob('gpurun_out/pmc_dec/*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'k_dec' not in k: continue
        tot.setdefault(k,{})
        tot[k][r['Counter_Name']]=tot[k].get(r['Counter_Name'],0.0)+float(r['Counter_Value'])
json.dump({"workload":"tools/dec_file.py decode: 4.0 GB x 200 columns of text, two decodes (sums over both)","kernels":tot}, open('gpurun_out/pmc_dec/summary.json','w'), indent=1, sort_keys=True)
for k,v in tot.items():
    print(k, {c:'%.3g'%x for c,x in sorted(v.items())})
PY
