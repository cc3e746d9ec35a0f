#!/bin/bash
mkdir -p gpurun_out/r4h
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4h/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4h/tests.log | tail -14
timeout -k 10 300 python bench.py > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err; tail -1 gpurun_out/r4h/bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], 'resident', d['ms_per_step_resident_inputs'], 'host issue', d['host_issue_ms_per_step'], d['inputs']); print(d['cpu_baseline'])"
grep cpu_baseline gpurun_out/r4h/bench.err | tail -12
