#!/bin/bash
mkdir -p gpurun_out/r4q
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4q/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4q/tests.log
