#!/bin/bash
mkdir -p gpurun_out/r4n
run() {
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --resident-inputs --clips-per-gpu 32 > gpurun_out/r4n/bench.log 2>&1; tail -1 gpurun_out/r4n/bench.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], [(b['kernel'], b['launches_per_step'], b['avg_us'], b['tflops']) for b in d['breakdown'] if 'wgrad' in b['kernel'] or b['kernel'].startswith('block_bwd')])"
}
AMT_WGRAD_BIG=0 run "small tiles, depth 1"
AMT_WGRAD_BIG=0 AMT_HIP_LIB=scratch/variants/libamt_hip_d2.so run "small tiles, depth 2"
AMT_WGRAD_BIG=0 AMT_HIP_LIB=scratch/variants/libamt_hip_d3.so run "small tiles, depth 3"
AMT_WGRAD_BIG=0 AMT_WGRAD_TARGET_WGS=512 run "small tiles, target 512"
AMT_WGRAD_BIG=0 AMT_WGRAD_TARGET_WGS=768 run "small tiles, target 768"
for t in 128 192 256 512; do AMT_WGRAD_TARGET_BIG=$t run "big tiles, target $t"; done
