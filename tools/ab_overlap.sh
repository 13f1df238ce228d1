#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
export ZDW_LIBDIR=${GRAFT_REPO_ROOT:-$PWD}/zdw_amd/lib_exp
# A/B: row lengths beside the dictionary sort (second stream) or after it
set -e
mkdir -p gpurun_out
run() { local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 5 --no-cpu --no-e2e --no-verify > gpurun_out/abo_$label.log 2>&1
  echo "$label $(grep -o '"stage_ms": {[^}]*}' gpurun_out/abo_$label.log) $(grep -o '"step_ms": {[^}]*}' gpurun_out/abo_$label.log)"
}
run base A=1
run serial ZDW_NO_OVERLAP=1
run wgs2 ZDW_P2_SIZE_WGS=2
run wgs4 ZDW_P2_SIZE_WGS=4
run wgs6 ZDW_P2_SIZE_WGS=6
