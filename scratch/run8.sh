cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3h
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3h/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3h/tests.log
timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3h/infer.json 2> gpurun_out/r3h/infer.txt; echo infer rc=$?
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3h/bench.json 2> gpurun_out/r3h/bench.txt; echo bench rc=$?
python - <<'PY'
import json
for f in ('infer','bench'):
    d=json.loads(open(f'gpurun_out/r3h/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], [(b['kernel'], b['avg_us']) for b in d['breakdown'][:9]])
PY
