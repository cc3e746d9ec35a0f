#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4d
python -m pytest tests -m gpu -x -q -s > gpurun_out/r4d/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4d/tests.log | grep -v "^\.\|warn\|Warn" | tail -25
for extra in "" "--force-overlap" "--force-overlap --no-graph"; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --profile-steps 0 $extra > gpurun_out/r4d/bench.log 2>&1; tail -1 gpurun_out/r4d/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$extra', d['ms_per_step'], 'host issue', d['host_issue_ms_per_step'])"
done
