mkdir -p gpurun_out/r03
export ZDW_LIBDIR=$GRAFT_REPO_ROOT/zdw_amd/lib_ab_w7x
for w in 20 24 28; do
  ZDW_P1_WAVES_PER_CU=$w timeout -k 10 200 python bench.py --steps 4 --warmup 2 --pass1-only --no-cpu --no-e2e --no-verify > gpurun_out/r03/w7x_$w.log 2>&1
  echo "w7 build, waves/CU $w: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/w7x_$w.log)"
done
timeout -k 10 200 python bench.py --steps 4 --warmup 2 --pass1-only --no-cpu --no-e2e --no-verify > gpurun_out/r03/w7x_auto.log 2>&1
echo "w7 build, auto: $(grep -o '"pass1": [0-9.]*' gpurun_out/r03/w7x_auto.log)"
