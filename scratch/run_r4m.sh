#!/bin/bash
mkdir -p gpurun_out/r4m
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "wgrad" > gpurun_out/r4m/tests_ops.log 2>&1; echo "ops rc=$?"; tail -4 gpurun_out/r4m/tests_ops.log
for big in 0 8192; do
AMT_WGRAD_BIG=$big timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --resident-inputs --clips-per-gpu 32 > gpurun_out/r4m/bench_$big.log 2>&1; tail -1 gpurun_out/r4m/bench_$big.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('WGRAD_BIG=$big', d['ms_per_step'])
for b in d['breakdown'][:8]: print('   ', b['kernel'], b['launches_per_step'], b['avg_us'], b['tflops'])"
done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4m/tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r4m/tests.log
