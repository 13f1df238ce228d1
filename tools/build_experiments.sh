#!/bin/bash
# A second copy of the library with the ablation hooks of the profiling tools compiled in (-DZDW_EXPERIMENTS; the shipped build has
# none): zdw_amd/lib_exp/.  Use it with ZDW_LIBDIR=$PWD/zdw_amd/lib_exp python bench.py ... and ZDW_DEBUG_SKIP=<bits>
# (1 no field work, 2 no dictionary probes, 4 no record stores, 32 the tokeniser alone: NOT valid encodes).
set -e
here=$(cd "$(dirname "$0")/.." && pwd)
make -C "$here/zdw_amd/csrc" -j4 lib OUT="$here/zdw_amd/lib_exp" EXTRA=-DZDW_EXPERIMENTS
make -C "$here/zdw_amd/csrc" OUT="$here/zdw_amd/lib_exp" "$here/zdw_amd/lib_exp/libzdw_synth.so"
echo "built $here/zdw_amd/lib_exp/libzdw_hip.so"
