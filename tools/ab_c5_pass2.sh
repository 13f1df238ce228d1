#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
# Pass 2's emit kernel on the C5 shape (1000 columns, every 20th of ids; 8 GB): workgroups per CU of its persistent grid
mkdir -p gpurun_out/p2
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
B="timeout -k 10 200 python bench.py --gb 8 --cols 1000 --id-every 20 --steps 3 --warmup 1 --no-cpu --no-e2e --no-verify"
run() { local label=$1; shift
  env "$@" $B > gpurun_out/p2/$label.log 2>&1
  echo "$label $(grep -o '"stage_ms": {[^}]*}' gpurun_out/p2/$label.log)"
}
run base A=1
for w in 2 3 4 5 6 7 8 10 12 16; do run wgs$w ZDW_P2_WGS=$w; done
