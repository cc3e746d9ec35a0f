#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ab12
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --workload infer --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/ab12/$name.json 2> gpurun_out/ab12/$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab12/$name.json").read().strip().splitlines()[-1]); print("infer $name", d["ms_per_step"], d["value"])
    for e in d.get("breakdown", []):
        if "qkv" in e["kernel"]: print("    ", e["kernel"], e["launches_per_step"], e["avg_us"])
except Exception as e: print("$name ERR", e)
PY
}
run base AMT_QKV_BIG=0
run qkvbig AMT_X=0
run base2 AMT_QKV_BIG=0
