#!/bin/bash
# pass 1 on the C2 shape (movie_tickets x 128: 67 M rows of 6 short fields, one new string per row), by part: experiments build,
# ZDW_DEBUG_SKIP = 32 tokeniser alone, 1 no field work, 2 no dictionary probes / inserts, 4 no record stores, 0 everything
mkdir -p gpurun_out/c2ab
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
for skip in 0 32 1 2 4; do
  ZDW_DEBUG_SKIP=$skip timeout -k 5 200 python tools/run_c2.py --copies 128 --steps 3 --pass1-only > gpurun_out/c2ab/skip_$skip.json 2> gpurun_out/c2ab/skip_$skip.err
  echo "skip $skip: $(tail -1 gpurun_out/c2ab/skip_$skip.json)"
done
