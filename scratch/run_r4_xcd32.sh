#!/bin/bash
O=gpurun_out/xcd32; mkdir -p $O
for T in 768 1120 1536; do
  AMT_WGRAD_XCD=2 AMT_WGRAD_TARGET_LARGE=$T timeout -k 10 300 python bench.py --clips-per-gpu 32 --resident-inputs --no-cpu-baseline --steps 20 --warmup 5 > $O/t$T.json 2> $O/t$T.err
  python - <<PY
import json
d=json.loads(open("$O/t$T.json").read().strip().splitlines()[-1]); print("xcd 2 target $T", d["ms_per_step"], [(b["kernel"],b["avg_us"]) for b in d["breakdown"] if "gemm_tn" in b["kernel"]])
PY
done
