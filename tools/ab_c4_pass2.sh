#!/bin/bash
# (the switches used here exist only in the experiments build: tools/build_experiments.sh)
# The benchmark's table: workgroups per CU of pass 2's emit kernel, and of the row-length kernel beside the dictionary sort
mkdir -p gpurun_out/p2
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
B="timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-verify"
run() { local label=$1; shift
  env "$@" $B > gpurun_out/p2/c4_$label.log 2>&1
  echo "$label $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/p2/c4_$label.log) $(grep -o '"dict_sort": [0-9.]*, "dict_emit": [0-9.]*, "pass2_size": [0-9.]*, "pass2_emit": [0-9.]*' gpurun_out/p2/c4_$label.log)"
}
for rep in 1 2; do
run base A=1
for w in 3 4 5 6; do run size$w ZDW_P2_SIZE_WGS=$w; done
done
