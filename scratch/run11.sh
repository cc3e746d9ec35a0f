cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3k/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3k/tests.log
run() { name=$1; shift; timeout -k 10 200 python bench.py "$@" > gpurun_out/r3k/$name.json 2> gpurun_out/r3k/$name.txt; echo $name rc=$?; }
run b8 --steps 20 --warmup 5 --no-cpu-baseline
run b8_fp8 --steps 20 --warmup 5 --no-cpu-baseline --attn-fp8
run b32 --steps 10 --warmup 3 --no-cpu-baseline --clips-per-gpu 32
run b32_fp8 --steps 10 --warmup 3 --no-cpu-baseline --clips-per-gpu 32 --attn-fp8
python - <<'PY'
import json
for f in ('b8','b8_fp8','b32','b32_fp8'):
    d=json.loads(open(f'gpurun_out/r3k/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['loss'], [(b['kernel'], b['avg_us']) for b in d['breakdown'] if 'attn_global' in b['kernel']])
PY
