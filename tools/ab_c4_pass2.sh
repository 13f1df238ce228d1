#!/bin/bash
mkdir -p gpurun_out/p2
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
B="timeout -k 10 200 python bench.py --steps 4 --warmup 1 --no-cpu --no-e2e --no-verify"
run() { local label=$1; shift
  env "$@" $B > gpurun_out/p2/c4_$label.log 2>&1
  echo "$label $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/p2/c4_$label.log) $(grep -o '"pass2_size": [0-9.]*, "pass2_emit": [0-9.]*' gpurun_out/p2/c4_$label.log)"
}
run base A=1
for w in 4 5 6 7 8 9 10 12 16; do run wgs$w ZDW_P2_WGS=$w; done
for w in 2 3 4 5 6 8; do run size$w ZDW_P2_SIZE_WGS=$w; done
