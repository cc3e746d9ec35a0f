#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4b
python -m pytest tests -m gpu -x -q > gpurun_out/r4b/tests.log 2>&1; echo "tests rc=$?"
grep -v "^$" gpurun_out/r4b/tests.log | tail -25
python scratch/parity_probe.py > gpurun_out/r4b/parity_probe.log 2>&1; grep "^gamma\|^events\|margin" gpurun_out/r4b/parity_probe.log | cut -c1-200
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4b/bench.log 2>&1; tail -1 gpurun_out/r4b/bench.log | cut -c1-400
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --force-overlap > gpurun_out/r4b/bench_overlap.log 2>&1; tail -1 gpurun_out/r4b/bench_overlap.log | cut -c1-400
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fused-optimizer --force-overlap > gpurun_out/r4b/bench_overlap_legacy.log 2>&1; tail -1 gpurun_out/r4b/bench_overlap_legacy.log | cut -c1-400
