#!/bin/bash
# The three sparse-table shapes of profiles/r02_sparse_2086cols.json (not the headline configuration): bench lines into gpurun_out/sparse_*.log
set -e
mkdir -p gpurun_out
timeout -k 10 200 python bench.py --steps 3 --gb 4 --cols 2086 --p-empty 0.97 --no-cpu --no-e2e > gpurun_out/sparse_97.log 2>&1
timeout -k 10 200 python bench.py --steps 3 --gb 4 --cols 2086 --p-empty 0.9 --no-cpu --no-e2e > gpurun_out/sparse_90.log 2>&1
timeout -k 10 200 python bench.py --steps 3 --gb 4 --cols 2086 --p-empty 0.75 --dead-cols 0.877 --no-cpu --no-e2e > gpurun_out/sparse_hits.log 2>&1
ZDW_P2_DENSE=1 timeout -k 10 200 python bench.py --steps 3 --gb 4 --cols 2086 --p-empty 0.75 --dead-cols 0.877 --no-cpu --no-e2e > gpurun_out/sparse_hits_dense.log 2>&1
ZDW_P1_NO_NZ=1 timeout -k 10 200 python bench.py --steps 3 --gb 4 --cols 2086 --p-empty 0.75 --dead-cols 0.877 --no-cpu --no-e2e > gpurun_out/sparse_hits_nonz.log 2>&1
for f in 97 90 hits hits_dense hits_nonz; do echo $f; grep -o '"value": [0-9.]*' gpurun_out/sparse_$f.log | head -n 1; grep -o '"stage_ms": {[^}]*}' gpurun_out/sparse_$f.log; done
