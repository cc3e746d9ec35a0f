#!/bin/bash
mkdir -p gpurun_out/r4l
for nc in 0 192; do
AMT_NOCHAIN=$nc timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --resident-inputs --clips-per-gpu 32 > gpurun_out/r4l/bench_$nc.log 2>&1; tail -1 gpurun_out/r4l/bench_$nc.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('NOCHAIN=$nc', d['ms_per_step'])
for b in d['breakdown'][:16]: print('   ', b['kernel'], b['launches_per_step'], b['avg_us'], b['tflops'])"
done
