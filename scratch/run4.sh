cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 5 120 ./scratch/big_bench 112000 0 | tail -1
timeout -k 5 120 ./scratch/big_bench 16000 1 | tail -1
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3d/tests.log 2>&1; echo tests rc=$?
tail -4 gpurun_out/r3d/tests.log
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3d/bench.json 2> gpurun_out/r3d/bench.txt; echo bench rc=$?
timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3d/infer.json 2> gpurun_out/r3d/infer.txt; echo infer rc=$?
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline > gpurun_out/r3d/b32.json 2> gpurun_out/r3d/b32.txt; echo b32 rc=$?
python - <<'PY'
import json
for f in ('bench','infer','b32'):
    d=json.loads(open(f'gpurun_out/r3d/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'])
    for b in d['breakdown'][:9]: print('   ', b['kernel'], b['launches_per_step'], b['avg_us'])
PY
