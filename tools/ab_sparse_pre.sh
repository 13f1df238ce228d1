#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
export ZDW_LIBDIR=${GRAFT_REPO_ROOT:-$PWD}/zdw_amd/lib_exp
set -e
mkdir -p gpurun_out
for v in pre nopre; do
  if [ $v = nopre ]; then export ZDW_P2S_NO_PRE=1; fi
  timeout -k 10 200 python bench.py --steps 3 --gb 4 --cols 2086 --p-empty 0.97 --no-cpu --no-e2e > gpurun_out/abp_$v.log 2>&1
  echo "$v $(grep -o '"value": [0-9.]*' gpurun_out/abp_$v.log | head -n 1) $(grep -o '"stage_ms": {[^}]*}' gpurun_out/abp_$v.log) $(grep -o '"verified": [a-z]*' gpurun_out/abp_$v.log)"
done
