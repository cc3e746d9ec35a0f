#!/bin/bash
mkdir -p gpurun_out/r4p
timeout -k 10 900 python -m pytest tests -m gpu -x -q -s > gpurun_out/r4p/tests.log 2>&1; echo "tests rc=$?"; grep -h "bf16 events" gpurun_out/r4p/tests.log; tail -3 gpurun_out/r4p/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --clips-per-gpu 32 > gpurun_out/r4p/bench_b32.json 2> gpurun_out/r4p/bench_b32.err; tail -1 gpurun_out/r4p/bench_b32.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('32 clips', d['ms_per_step'], d['value'], 'resident', d['ms_per_step_resident_inputs'])"
timeout -k 10 300 python bench.py --workload infer --no-cpu-baseline > gpurun_out/r4p/bench_infer.json 2> gpurun_out/r4p/bench_infer.err; tail -1 gpurun_out/r4p/bench_infer.json | cut -c1-600
