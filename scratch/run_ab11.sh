#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ab11
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab11/$name.json 2> gpurun_out/ab11/$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab11/$name.json").read().strip().splitlines()[-1]); print("$name", d["ms_per_step"], d["value"], d.get("loss"))
    for e in d.get("breakdown", []):
        if e["kernel"] in ("ffn_fwd_fused","qkv_fwd_fused","ffn_qkv_fwd_fused"): print("    ", e["kernel"], e["launches_per_step"], e["avg_us"])
except Exception as e: print("$name ERR", e)
PY
}
run base AMT_X=0
run nob_w4 AMT_BOUNDARY=0
run nob_w8 AMT_BOUNDARY=0 AMT_FFN_W8=1
run nob_w4b AMT_BOUNDARY=0
