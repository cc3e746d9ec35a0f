#!/bin/bash
O=gpurun_out/late32; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --clips-per-gpu 32 --resident-inputs --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); print("$name", d["ms_per_step"], [(b["kernel"],b["launches_per_step"],b["avg_us"]) for b in d["breakdown"] if "gemm_tn" in b["kernel"]])
except Exception as e: print("$name ERR", e)
PY
}
run g8 AMT_LATE_WGRAD_GROUP=8
run g16 AMT_LATE_WGRAD_GROUP=16
run g16_768 AMT_LATE_WGRAD_GROUP=16 AMT_LATE_WGRAD_WGS=768
run g16_1024 AMT_LATE_WGRAD_GROUP=16 AMT_LATE_WGRAD_WGS=1024
run g12 AMT_LATE_WGRAD_GROUP=12
