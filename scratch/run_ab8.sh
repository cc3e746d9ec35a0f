#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ab8
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --clips-per-gpu 32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab8/b32_$name.json 2> gpurun_out/ab8/b32_$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab8/b32_$name.json").read().strip().splitlines()[-1]); print("b32 $name", d["ms_per_step"], d["value"], d["loss"])
except Exception as e: print("b32 $name ERR", e)
PY
}
run base AMT_X=0
run c64 AMT_BB_LONG=1
run c128 AMT_BB_LONG=2
run c256 AMT_BB_LONG=4
run all AMT_BB_LONG=7
