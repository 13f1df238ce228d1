#!/bin/bash
# round 3 baseline on this round's box: the default bench line, the separate-row-scan form, and a few pass-1 knobs
mkdir -p gpurun_out/r03
run() { label=$1; shift; env "$@" timeout -k 10 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/r03/base_$label.log 2>&1; echo "$label rc=$? $(grep -o '"stage_ms": {[^}]*}' gpurun_out/r03/base_$label.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03/base_$label.log)"; }
run fold A=1
run nofold ZDW_P1_NO_FOLD=1
run slots24 ZDW_DICT_SLOTS=16777216
run w20 ZDW_P1_WAVES_PER_CU=20
