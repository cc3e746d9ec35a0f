#!/bin/bash
mkdir -p gpurun_out/r4f
python -m pytest tests -m gpu -x -q -k "in_place or two_phase or fused or one_launch or data_parallel" > gpurun_out/r4f/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4f/tests.log | tail -12
for i in 1 2; do
for mode in "" "--force-overlap" "--force-overlap --no-signal" "NOPG"; do
extra="$mode"
if [ "$mode" = NOPG ]; then extra="--force-overlap"; export AMT_BENCH_NO_PG=1; else unset AMT_BENCH_NO_PG; fi
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --profile-steps 0 $extra > gpurun_out/r4f/bench.log 2>&1; tail -1 gpurun_out/r4f/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$mode', d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'], d['config']['grad_allreduce'])"
done; done
