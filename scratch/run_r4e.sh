#!/bin/bash
mkdir -p gpurun_out/r4e
for i in 1 2; do
for mode in "" "--force-overlap" "NOPG"; do
extra="$mode"; env_pg=""
if [ "$mode" = NOPG ]; then extra="--force-overlap"; export AMT_BENCH_NO_PG=1; else unset AMT_BENCH_NO_PG; fi
python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 $extra > gpurun_out/r4e/bench.log 2>&1; tail -1 gpurun_out/r4e/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])"
done; done
