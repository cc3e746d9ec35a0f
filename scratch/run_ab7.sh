#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ab7
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --workload infer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab7/infer_$name.json 2> gpurun_out/ab7/infer_$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab7/infer_$name.json").read().strip().splitlines()[-1]); print("infer $name", d["ms_per_step"], d["value"])
except Exception as e: print("infer $name ERR", e)
PY
}
run base AMT_X=0
run c16 AMT_BLOCK_OCC2=1
run c64 AMT_BLOCK_OCC2=2
run c128 AMT_BLOCK_OCC2=4
run c256 AMT_BLOCK_OCC2=8
run all AMT_BLOCK_OCC2=15 AMT_BIG=5
run all_but256 AMT_BLOCK_OCC2=7 AMT_BIG=5
