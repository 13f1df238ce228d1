#!/bin/bash
# pass 1 against the number of resident waves per CU (experiments build: tools/build_experiments.sh): T(w) says what a 7th and 8th wave per SIMD would buy
mkdir -p gpurun_out/r03
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
for w in 8 12 16 20 24; do
  ZDW_P1_WAVES_PER_CU=$w timeout -k 10 200 python bench.py --steps 4 --warmup 2 --pass1-only --no-cpu --no-e2e --no-verify > gpurun_out/r03/waves_$w.log 2>&1
  echo "waves/CU $w: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/waves_$w.log)"
done
