#!/bin/bash
mkdir -p gpurun_out/r4r
for g in 1.0 1e-6; do timeout -k 10 300 python probes/bf16_layer_probe.py --gamma $g > gpurun_out/r4r/layer_probe_$g.txt 2>&1; tail -3 gpurun_out/r4r/layer_probe_$g.txt; done
timeout -k 10 300 python probes/bf16_layer_probe.py --gamma 1.0 --dtype float32 > gpurun_out/r4r/layer_probe_f32.txt 2>&1; tail -2 gpurun_out/r4r/layer_probe_f32.txt
