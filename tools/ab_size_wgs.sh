#!/bin/bash
# the row-length kernel's workgroups per CU beside the dictionary sort (experiments build): step time and the two stages
mkdir -p gpurun_out/r03
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
for w in 2 3 4 5 6; do
  ZDW_P2_SIZE_WGS=$w timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/r03/sizewgs_$w.log 2>&1
  echo "size WGs/CU $w: $(grep -o '"dict_sort": [0-9.]*' gpurun_out/r03/sizewgs_$w.log) $(grep -o '"pass2_size": [0-9.]*' gpurun_out/r03/sizewgs_$w.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03/sizewgs_$w.log) $(grep -o '"median": [0-9.]*' gpurun_out/r03/sizewgs_$w.log)"
done
