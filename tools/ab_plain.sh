#!/bin/bash
# A/B: the first table probe as an ordinary load (this build) with the image streamed by non-temporal loads (lib) or ordinary ones (lib_exp, -DZDW_P1_PLAIN_STREAM)
mkdir -p gpurun_out/r03
run() { label=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/r03/plain_$label.log 2>&1; echo "$label: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/plain_$label.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03/plain_$label.log)"; }
run nt A=1
run plainstream ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_exp
run nt_again A=1
