#!/bin/bash
AMT_WGRAD_BIG=0 timeout -k 10 120 python scratch/wgrad_big_bench.py 2>&1 | grep -v amdgpu
for t in 256 512; do AMT_WGRAD_TARGET_BIG=$t timeout -k 10 120 python scratch/wgrad_big_bench.py 2>&1 | grep -v amdgpu; done
