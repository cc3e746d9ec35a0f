#!/bin/bash
O=gpurun_out/late32; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 300 python bench.py --clips-per-gpu 32 --resident-inputs --no-cpu-baseline --steps 20 --warmup 5 > $O/$name.json 2> $O/$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$O/$name.json").read().strip().splitlines()[-1]); print("$name", d["ms_per_step"], [(b["kernel"],b["launches_per_step"],b["avg_us"]) for b in d["breakdown"] if "gemm_tn" in b["kernel"] or "block_bwd" in b["kernel"]])
except Exception as e: print("$name ERR", e)
PY
}
run base AMT_X=0
run late_all_1024 AMT_CNN_WGRAD_LATE=1
run late_all_512 AMT_CNN_WGRAD_LATE=1 AMT_LATE_WGRAD_WGS=512
run late_all_2048 AMT_CNN_WGRAD_LATE=1 AMT_LATE_WGRAD_WGS=2048
run late_c128_1024 AMT_CNN_WGRAD_LATE=128
