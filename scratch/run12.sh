cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3l
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3l/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3l/tests.log
run() { name=$1; shift; timeout -k 10 200 python bench.py "$@" > gpurun_out/r3l/$name.json 2> gpurun_out/r3l/$name.txt; echo $name rc=$?; }
run b8 --steps 20 --warmup 5 --no-cpu-baseline
run b8_unfused --steps 20 --warmup 5 --no-cpu-baseline --no-fused-optimizer
python - <<'PY'
import json
for f in ('b8','b8_unfused'):
    d=json.loads(open(f'gpurun_out/r3l/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['loss'])
PY
grep -E "adamw|convert|finalize|reduce_part" gpurun_out/r3l/b8.txt gpurun_out/r3l/b8_unfused.txt
