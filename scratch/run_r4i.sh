#!/bin/bash
mkdir -p gpurun_out/r4i
cat /sys/fs/cgroup/cpu.max 2>&1 | head -2; nproc
timeout -k 10 400 python bench.py > gpurun_out/r4i/bench.json 2> gpurun_out/r4i/bench.err; tail -1 gpurun_out/r4i/bench.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], 'resident', d['ms_per_step_resident_inputs'], 'host issue', d['host_issue_ms_per_step'], d['inputs']); print(d['cpu_baseline'])"
grep cpu_baseline gpurun_out/r4i/bench.err | tail -14
timeout -k 10 300 python bench.py --clips-per-gpu 32 --no-cpu-baseline > gpurun_out/r4i/bench_b32.json 2> gpurun_out/r4i/bench_b32.err; tail -1 gpurun_out/r4i/bench_b32.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('32 clips', d['ms_per_step'], d['value'], 'resident', d['ms_per_step_resident_inputs'])
for b in d['breakdown']: print(b)"
