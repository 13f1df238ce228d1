#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
export ZDW_LIBDIR=${GRAFT_REPO_ROOT:-$PWD}/zdw_amd/lib_exp
# A/B: batches of 64 columns per wave in the decoder's row kernel (ZDW_DEC_GROUP), 10 GB x 200 columns
set -e
mkdir -p gpurun_out
for g in 1 2 4; do
  ZDW_DEC_GROUP=$g timeout -k 10 300 python tools/run_decode.py --gb 10 --out gpurun_out/dec_g$g.json > gpurun_out/dec_g$g.log 2>&1
  echo "group $g: $(tail -n 1 gpurun_out/dec_g$g.log)"
done
