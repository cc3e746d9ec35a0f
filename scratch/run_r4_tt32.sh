#!/bin/bash
O=gpurun_out/tt32; mkdir -p $O
for X in 16 32; do
  AMT_FFN_BWD_TT=$X timeout -k 10 300 python bench.py --clips-per-gpu 32 --resident-inputs --no-cpu-baseline --steps 20 --warmup 5 > $O/x$X.json 2> $O/x$X.err
  python - <<PY
import json
d=json.loads(open("$O/x$X.json").read().strip().splitlines()[-1]); print("ffn_bwd tt $X", d["ms_per_step"], [(b["kernel"],b["avg_us"]) for b in d["breakdown"] if "ffn_bwd" in b["kernel"]])
PY
done
