#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ab9
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab9/cfg2_$name.json 2> gpurun_out/ab9/cfg2_$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab9/cfg2_$name.json").read().strip().splitlines()[-1]); print("cfg2 $name", d["ms_per_step"], d["value"], d["loss"])
except Exception as e: print("cfg2 $name ERR", e)
PY
}
for v in "$@"; do
  if [ $v = base ]; then run base AMT_X=0; else run $v AMT_HIP_LIB=$PWD/scratch/variants/libamt_hip_$v.so; fi
done
run base2 AMT_X=0
