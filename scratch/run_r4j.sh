#!/bin/bash
mkdir -p gpurun_out/r4j
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "in_place or two_phase or fused or one_launch or data_parallel or train_entry" > gpurun_out/r4j/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4j/tests.log | tail -8
for mode in "" "--force-overlap" "--force-overlap --no-signal" "--resident-inputs" "--resident-inputs --force-overlap" "--resident-inputs --force-overlap --no-signal"; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 $mode > gpurun_out/r4j/bench.log 2>&1; tail -1 gpurun_out/r4j/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'resident', d['ms_per_step_resident_inputs'], 'host issue', d['host_issue_ms_per_step'])"
done
