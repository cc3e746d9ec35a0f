#!/bin/bash
mkdir -p gpurun_out/r4k
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4k/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4k/tests.log | tail -12
for mode in "--resident-inputs" ""; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline $mode > gpurun_out/r4k/bench.log 2>&1; tail -1 gpurun_out/r4k/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'resident', d['ms_per_step_resident_inputs'])
for b in d['breakdown'][:14]: print('   ', b['kernel'], b['launches_per_step'], b['avg_us'])"
done
