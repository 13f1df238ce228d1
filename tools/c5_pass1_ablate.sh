#!/bin/bash
# pass 1 on the C5 shape (1000 columns, every 20th a column of practically unique 24-character ids), 4 GB, by part: experiments build,
# ZDW_DEBUG_SKIP = 32 tokeniser alone, 1 no field work, 2 no dictionary probes / inserts, 4 no record stores, 0 everything
mkdir -p gpurun_out/c5ab
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
for skip in 0 32 1 2 4; do
  ZDW_DEBUG_SKIP=$skip timeout -k 5 200 python bench.py --gb 4 --cols 1000 --id-every 20 --steps 3 --warmup 1 --pass1-only --no-cpu --no-e2e --no-verify > gpurun_out/c5ab/c5ab_$skip.json 2> gpurun_out/c5ab/c5ab_$skip.err
  echo "skip $skip: $(grep -o '"pass1": [0-9.]*' gpurun_out/c5ab/c5ab_$skip.json)"
done
