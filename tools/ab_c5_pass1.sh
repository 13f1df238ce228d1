#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
export ZDW_LIBDIR=${GRAFT_REPO_ROOT:-$PWD}/zdw_amd/lib_exp
# What pass 1 costs on C5's table shape (4 GB, one block, no memory model): columns and id columns varied, field work ablated
set -e
mkdir -p gpurun_out
run() { local label=$1; shift
  env "$@" > gpurun_out/c5p_$label.log 2>&1 || true
  echo "$label $(grep -o '"pass1": [0-9.]*' gpurun_out/c5p_$label.log | head -n 1) $(grep -o '"dict_entries": [0-9]*' gpurun_out/c5p_$label.log | head -n 1)"
}
B="timeout -k 10 200 python bench.py --steps 2 --gb 4 --no-cpu --no-e2e --no-verify --pass1-only"
run c200 $B
run c1000 $B --cols 1000
run c1000_ids $B --cols 1000 --id-every 20
run c200_ids $B --cols 200 --id-every 20
run c1000_ids_nofield ZDW_DEBUG_SKIP=1 $B --cols 1000 --id-every 20
run c1000_ids_noprobe ZDW_DEBUG_SKIP=2 $B --cols 1000 --id-every 20
