#!/bin/bash
# rocprofv3 kernel trace + stats of the sparse-table run (2086 columns, 97 % empty, 4 GB): profiles/r02_sparse_97_kernel_stats.csv
set -e
out=gpurun_out/prof_sparse
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 3 --warmup 1 --gb 4 --cols 2086 --p-empty 0.97 --no-cpu --no-e2e --no-verify > $out/bench_trace.json 2> $out/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --steps 1 --warmup 0 --gb 4 --cols 2086 --p-empty 0.97 --no-cpu --no-e2e --no-verify > $out/bench_fetch.json 2> $out/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --steps 1 --warmup 0 --gb 4 --cols 2086 --p-empty 0.97 --no-cpu --no-e2e --no-verify > $out/bench_write.json 2> $out/bench_write.err
find $out -name "*.csv" | head -20
