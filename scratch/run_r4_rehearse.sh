#!/bin/bash
# the driver's N > 1 launch line, rehearsed on the one-GPU box: ranks share cuda:0 and exchange through gloo
set -e
O=gpurun_out/rehearse; mkdir -p $O
for N in 2 4; do
  AMT_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) \
    bench.py --gpus $N --steps 10 --warmup 3 > $O/n$N.json 2> $O/n$N.err
  tail -c 1500 $O/n$N.json
done
