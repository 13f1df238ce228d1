#!/bin/bash
# A/B: how many descriptors pass 1 leaves pending behind a chunk A / a round (k_pass1.hip: P1_KEEP_AFTER_A / _B).  The variants are
# built beforehand: make -C zdw_amd/csrc lib OUT=zdw_amd/lib_keep_A_B EXTRA="-DP1_KEEP_AFTER_A=Au -DP1_KEEP_AFTER_B=Bu"
mkdir -p gpurun_out/r03
for d in zdw_amd/lib zdw_amd/lib_keep_*; do
  ZDW_LIBDIR=$GRAFT_REPO_ROOT/$d timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/r03/keep_$(basename $d).log 2>&1
  echo "$d: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/keep_$(basename $d).log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03/keep_$(basename $d).log)"
done
