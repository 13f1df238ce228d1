#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun): kernel trace + stats, then PMC
# passes for HBM traffic (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, 3+2).
set -e
tag=${1:-r01}
out=gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 3 --warmup 1 --no-cpu --no-e2e --no-verify > $out/bench_trace.json 2> $out/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify > $out/bench_fetch.json 2> $out/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- python bench.py --steps 1 --warmup 0 --no-cpu --no-e2e --no-verify > $out/bench_write.json 2> $out/bench_write.err
find $out -name "*.csv" | head -20
