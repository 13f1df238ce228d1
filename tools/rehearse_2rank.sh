#!/bin/bash
# 2 (or $1) ranks on ONE GPU over gloo: exercises bench.py's distributed path (sharded.encode_shard_dist) end to end.
export HSA_ENABLE_IPC_MODE_LEGACY=0
n=${1:-2}
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 \
   bench.py --gpus $n --steps 2 --warmup 1 --gb ${2:-0.5} --no-cpu --backend gloo --one-device $3
