#!/bin/bash
# round 4, first GPU pass: op tests + model tests + parity probe after the residual-stream storage change
set -o pipefail
mkdir -p gpurun_out/r4a
python -m pytest tests -m gpu -x -q > gpurun_out/r4a/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/r4a/tests.log
tail -15 gpurun_out/r4a/tests.log
python scratch/parity_probe.py > gpurun_out/r4a/parity_probe.log 2>&1; tail -30 gpurun_out/r4a/parity_probe.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r4a/bench.log 2>&1; tail -3 gpurun_out/r4a/bench.log
