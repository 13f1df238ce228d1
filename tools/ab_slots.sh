#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
export ZDW_LIBDIR=${GRAFT_REPO_ROOT:-$PWD}/zdw_amd/lib_exp
# A/B: size of the first dictionary table on the benchmark's table (10 GB): 2^25 (default cap) / 2^26 / 2^27 slots of 32 bytes
set -e
mkdir -p gpurun_out
for n in 0 67108864 134217728; do
  if [ $n != 0 ]; then export ZDW_DICT_SLOTS=$n; fi
  timeout -k 10 200 python bench.py --steps 3 --no-cpu --no-e2e --no-verify > gpurun_out/abs_$n.log 2>&1
  echo "slots $n: $(grep -o '"pass1": [0-9.]*' gpurun_out/abs_$n.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/abs_$n.log)"
done
