# FETCH_SIZE / WRITE_SIZE calibration (scratch/fetch_calib.hip): gpurun -- 'bash scratch/fetch_calib.sh > gpurun_out/fetch_calib.txt'
set -u
R=$GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib $R/scratch/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fc_fetch /tmp/fc_write
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/fc_fetch -- /tmp/fetch_calib > /tmp/fc_fetch.log 2>&1 || { tail -5 /tmp/fc_fetch.log; exit 1; }
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/fc_write -- /tmp/fetch_calib > /tmp/fc_write.log 2>&1 || { tail -5 /tmp/fc_write.log; exit 1; }
grep "known bytes" /tmp/fc_fetch.log
python3 - <<'PY'
import csv, glob
known = {"k_stream16": [2**30], "k_panel128": [2**30 // 512 * 128, 2**30 // 1024 * 128, 2**30 // 2048 * 128], "k_atomic64": [8 * 65536 * 64 * 4], "k_store16": [2**30]}
def load(d, counter):
    rows = []
    for r in csv.DictReader(open(glob.glob(d + "/*/*_counter_collection.csv")[0])):
        if r["Counter_Name"] == counter:
            rows.append((int(r.get("Dispatch_Id", 0) or 0), r["Kernel_Name"], float(r["Counter_Value"])))
    return sorted(rows)
fe, wr = load("/tmp/fc_fetch", "FETCH_SIZE"), load("/tmp/fc_write", "WRITE_SIZE")
print("%-14s %14s %16s %10s %16s %10s" % ("kernel", "known bytes", "FETCH_SIZE*1024", "/ known", "WRITE_SIZE*1024", "/ known"))
seen = {}
for (_, name, f), (_, _, w) in zip(fe, wr):
    base = name.split("(")[0]
    if base not in known:
        continue
    i = seen.get(base, 0); seen[base] = i + 1
    k = known[base][min(i, len(known[base]) - 1)]
    print("%-14s %14d %16.0f %10.3f %16.0f %10.3f" % (base + ("#%d" % i if len(known[base]) > 1 else ""), k, f * 1024, f * 1024 / k, w * 1024, w * 1024 / k))
PY
