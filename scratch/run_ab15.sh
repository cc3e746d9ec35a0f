#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/ab15
run() { # name, args, env...
  local name=$1; local args=$2; shift 2
  env "$@" python3 bench.py $args --no-cpu-baseline > gpurun_out/ab15/$name.json 2> gpurun_out/ab15/$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab15/$name.json").read().strip().splitlines()[-1]); print("$name", d["ms_per_step"], d["value"], d.get("loss"))
    for e in d.get("breakdown", []):
        if "attn_local" in e["kernel"] or "qkv_fwd" in e["kernel"]: print("      ", e["kernel"], e["launches_per_step"], e["avg_us"])
except Exception as e: print("$name ERR", e)
PY
}
run cfg2_new "--steps 20 --warmup 5" AMT_X=0
run cfg2_old "--steps 20 --warmup 5" AMT_PREROT_LOCAL=0
run cfg2_new2 "--steps 20 --warmup 5" AMT_X=0
run cfg2_old2 "--steps 20 --warmup 5" AMT_PREROT_LOCAL=0
run b32_new "--clips-per-gpu 32 --steps 10 --warmup 3" AMT_X=0
run b32_old "--clips-per-gpu 32 --steps 10 --warmup 3" AMT_PREROT_LOCAL=0
run infer_new "--workload infer --steps 10 --warmup 3" AMT_X=0
run infer_old "--workload infer --steps 10 --warmup 3" AMT_PREROT_LOCAL=0
