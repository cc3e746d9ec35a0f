#!/bin/bash
# per-launch durations of the grouped weight-gradient launches at 32 clips per GPU (eager steps)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o; rm -rf gpurun_out/r4o/kt
AMT_WGRAD_BIG=${BIG:-0} AMT_WGRAD_TARGET_WGS=${TGT:-512} rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r4o/kt -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --resident-inputs --clips-per-gpu 32 --no-graph --profile-steps 0 > gpurun_out/r4o/b.json 2> gpurun_out/r4o/b.err
f=$(find gpurun_out/r4o/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))))
rows.sort()
# last step: from the last stem_fwd on
idx = max(i for i, r in enumerate(rows) if "stem_fwd" in r[2])
seq = rows[idx:]
tot = 0
for s, e, n, wg in seq:
    if "gemm_tn_group" in n or "gemm_tn_kernel" in n:
        tot += (e - s) / 1e3
        nm = re.sub(r"\(.*", "", n)[:60]
        print(f"{(e - s) / 1e3:8.1f} us  {wg:5d} wgs  {nm}")
print("total", tot)
PY
rm -rf gpurun_out/r4o/kt
