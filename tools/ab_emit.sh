#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
export ZDW_LIBDIR=${GRAFT_REPO_ROOT:-$PWD}/zdw_amd/lib_exp
# A/B of the emit kernel's forms on the benchmark's table (stage_ms per variant)
set -e
mkdir -p gpurun_out
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 3 --no-cpu --no-e2e --no-verify > gpurun_out/ab_$label.log 2>&1
  echo "$label $(grep -o '"pass2_emit": [0-9.]*' gpurun_out/ab_$label.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab_$label.log)"
}
run base A=1
run nopipe7 ZDW_P2_EMIT_NOPIPE=1 ZDW_P2_WGS=7
run nopipe8 ZDW_P2_EMIT_NOPIPE=1 ZDW_P2_WGS=8
run rows4_8 ZDW_P2_EMIT_ROWS=4 ZDW_P2_WGS=8
run rows4_7 ZDW_P2_EMIT_ROWS=4 ZDW_P2_WGS=7
run rows16 ZDW_P2_EMIT_ROWS=16
