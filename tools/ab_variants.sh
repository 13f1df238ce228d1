#!/bin/bash
# Times pass 1 of the default bench with the shipped build and with every variant tools/build_variant.sh made.
mkdir -p gpurun_out/r03
for rep in 1 2; do
for d in zdw_amd/lib zdw_amd/lib_ab_*; do
  [ -d "$d" ] || continue
  ZDW_LIBDIR=$GRAFT_REPO_ROOT/$d timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/r03/ab_$(basename $d).log 2>&1
  echo "$d: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/ab_$(basename $d).log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03/ab_$(basename $d).log)"
done
done
