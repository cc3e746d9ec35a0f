#!/bin/bash
mkdir -p gpurun_out/r4s
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4s/tests.log 2>&1; echo "tests rc=$?"; tail -2 gpurun_out/r4s/tests.log
for g in 1.0 1e-6; do timeout -k 10 300 python probes/bf16_layer_probe.py --gamma $g 2>&1 | grep -v amdgpu > gpurun_out/r4s/layer_probe_$g.txt; tail -1 gpurun_out/r4s/layer_probe_$g.txt; done
timeout -k 10 300 python scratch/parity_probe.py > gpurun_out/r4s/parity_probe.log 2>&1; grep "^gamma" gpurun_out/r4s/parity_probe.log | cut -c1-160
