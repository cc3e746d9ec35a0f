#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
python -m pytest tests -m gpu -x -q > gpurun_out/r4c/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4c/tests.log | tail -12
for mode in fused overlap; do
  rm -rf gpurun_out/r4c/ks_$mode
  extra=""; [ $mode = overlap ] && extra="--force-overlap"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4c/ks_$mode -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 $extra > gpurun_out/r4c/ks_$mode.json 2> gpurun_out/r4c/ks_$mode.err
  f=$(find gpurun_out/r4c/ks_$mode -name "*kernel_trace.csv" | head -1)
  python3 scratch/trace_gaps.py $f > gpurun_out/r4c/gaps_$mode.txt 2>&1
  rm -rf gpurun_out/r4c/ks_$mode
  echo "== $mode"; tail -1 gpurun_out/r4c/ks_$mode.json | cut -c1-200; cat gpurun_out/r4c/gaps_$mode.txt
done
