#!/bin/bash
# Pass 1 with the shipped build and with every variant of tools/build_variant.sh: the benchmark's table, the C5 shape (1000 columns,
# every 20th a column of practically unique 24-character ids; 4 GB, one block, no memory model), the C2 shape (67 M rows, a new
# date-time in every row)
mkdir -p gpurun_out/ab
for rep in 1 2; do
for d in zdw_amd/lib zdw_amd/lib_ab_*; do
  [ -d "$d" ] || continue
  b=$(basename $d)
  ZDW_LIBDIR=$GRAFT_REPO_ROOT/$d timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/ab/c4_$b.log 2>&1 || exit 1
  ZDW_LIBDIR=$GRAFT_REPO_ROOT/$d timeout -k 10 200 python bench.py --gb 4 --cols 1000 --id-every 20 --steps 3 --warmup 1 --pass1-only --no-cpu --no-e2e --no-verify > gpurun_out/ab/c5_$b.log 2>&1 || exit 1
  ZDW_LIBDIR=$GRAFT_REPO_ROOT/$d timeout -k 10 200 python tools/run_c2.py --copies 128 --steps 3 --pass1-only > gpurun_out/ab/c2_$b.log 2>&1 || exit 1
  echo "$b: C4 $(grep -o '"pass1": [0-9.]*' gpurun_out/ab/c4_$b.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/ab/c4_$b.log) | C5 shape 4 GB $(grep -o '"pass1": [0-9.]*' gpurun_out/ab/c5_$b.log) | C2 shape $(tail -1 gpurun_out/ab/c2_$b.log | cut -c1-60)"
done
done
