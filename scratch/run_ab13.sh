#!/bin/bash
# usage: run_ab13.sh "<bench args>" variant...  ; prints step time and the attention classes of the breakdown
set -o pipefail
mkdir -p gpurun_out/ab13
args=$1; shift
run() {
  local name=$1; shift
  env "$@" python3 bench.py $args --no-cpu-baseline > gpurun_out/ab13/$name.json 2> gpurun_out/ab13/$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab13/$name.json").read().strip().splitlines()[-1]); print("$args | $name", d["ms_per_step"], d["value"], d.get("loss"))
    for e in d.get("breakdown", []):
        if "attn" in e["kernel"]: print("      ", e["kernel"], e["launches_per_step"], e["avg_us"])
except Exception as e: print("$name ERR", e)
PY
}
for v in "$@"; do
  if [ $v = base ]; then run base AMT_X=0; else run $v AMT_HIP_LIB=$PWD/scratch/variants/libamt_hip_$v.so; fi
done
