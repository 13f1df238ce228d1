#!/bin/bash
# pass 1 with 24 and 28 waves per CU in workgroups of 4 / 7 / 8 / 12 / 14 waves: needs the variant with registers for 7 waves per SIMD and
# the experiments' switches: tools/build_variant.sh w7x "-DP1_MIN_WAVES=7 -DZDW_EXPERIMENTS"
mkdir -p gpurun_out/r03
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_ab_w7x
run() { label=$1; shift; env "$@" timeout -k 10 200 python bench.py --steps 4 --warmup 2 --pass1-only --no-cpu --no-e2e --no-verify > gpurun_out/r03/w7b_$label.log 2>&1; echo "$label: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/w7b_$label.log)"; }
run wpb8_w24 ZDW_P1_WPB=8 ZDW_P1_WAVES_PER_CU=24
run wpb14_w28 ZDW_P1_WPB=14 ZDW_P1_WAVES_PER_CU=28
run wpb7_w28 ZDW_P1_WPB=7 ZDW_P1_WAVES_PER_CU=28
run wpb12_w24 ZDW_P1_WPB=12 ZDW_P1_WAVES_PER_CU=24
run wpb4_w24 ZDW_P1_WPB=4 ZDW_P1_WAVES_PER_CU=24
