#!/bin/bash
# usage: run_ab9.sh <bench args...> -- variant...   (variant "base" = the in-tree library)
set -o pipefail
mkdir -p gpurun_out/ab9
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done; shift
run() { # name, env...
  local name=$1; shift
  env "$@" python3 bench.py "${args[@]}" --no-cpu-baseline > gpurun_out/ab9/$name.json 2> gpurun_out/ab9/$name.err || echo "FAILED $name"
  python3 - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab9/$name.json").read().strip().splitlines()[-1]); print("${args[*]} | $name", d["ms_per_step"], d["value"], d.get("loss"))
except Exception as e: print("$name ERR", e)
PY
}
for v in "$@"; do
  if [ $v = base ]; then run base AMT_X=0; else run $v AMT_HIP_LIB=$PWD/scratch/variants/libamt_hip_$v.so; fi
done
