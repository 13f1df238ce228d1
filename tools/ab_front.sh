#!/bin/bash
# A/B: size of the front table of the dictionary (zdw_dict_dev.h) on the benchmark's table
mkdir -p gpurun_out/r03
for f in 0 16384 32768 65536 131072; do
  ZDW_FRONT_SLOTS=$f timeout -k 10 200 python bench.py --steps 4 --warmup 2 --no-cpu --no-e2e --no-verify > gpurun_out/r03/front_$f.log 2>&1
  echo "front $f: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/front_$f.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r03/front_$f.log)"
done
