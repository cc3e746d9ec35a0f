#!/bin/bash
mkdir -p gpurun_out/r4g
timeout -k 10 400 python scratch/overlap_cost.py 2>&1 | grep -v amdgpu.ids | tail -12
AMT_HALF_WAIT_KERNEL=1 timeout -k 10 400 python scratch/overlap_cost.py 2>&1 | grep -v amdgpu.ids | tail -6
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "in_place or two_phase or fused or one_launch or data_parallel" > gpurun_out/r4g/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4g/tests.log | tail -6
for mode in "" "--force-overlap" "--force-overlap --no-signal"; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 $mode > gpurun_out/r4g/bench.log 2>&1; tail -1 gpurun_out/r4g/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])"
done
