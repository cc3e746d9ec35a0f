cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3g/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3g/tests.log
timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3g/infer.json 2> gpurun_out/r3g/infer.txt; echo infer rc=$?
AMT_BLOCK_INF_LONG=0 timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3g/infer_short.json 2> gpurun_out/r3g/infer_short.txt; echo infer rc=$?
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3g/bench.json 2> gpurun_out/r3g/bench.txt; echo bench rc=$?
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline > gpurun_out/r3g/b32.json 2> gpurun_out/r3g/b32.txt; echo b32 rc=$?
python - <<'PY'
import json
for f in ('infer','infer_short','bench','b32'):
    d=json.loads(open(f'gpurun_out/r3g/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], [(b['kernel'], b['avg_us']) for b in d['breakdown'][:8]])
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3g/prof_infer -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --workload infer --steps 5 --warmup 2 --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r3g/prof_infer.log 2>&1; echo prof rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/r3g/prof_infer -name "*kernel_trace.csv" -delete; find gpurun_out/r3g/prof_infer -name "*.db" -delete
