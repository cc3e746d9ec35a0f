cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3j
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3j/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3j/tests.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py $BARGS > gpurun_out/r3j/$name.json 2> gpurun_out/r3j/$name.txt; echo $name rc=$?; }
BARGS="--steps 20 --warmup 5 --no-cpu-baseline"; run bench A=1
BARGS="--steps 20 --warmup 5 --no-cpu-baseline --no-fused-optimizer"; run bench_unfused A=1
BARGS="--steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline"; run b32 A=1
python - <<'PY'
import json,glob
for f in ('bench','bench_unfused','b32'):
    d=json.loads(open(f'gpurun_out/r3j/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], d['loss'], [(b['kernel'], b['avg_us']) for b in d['breakdown'] if b['kernel'] in ('adamw_norm','adamw_commit','adamw_compute','adamw_apply','convert_batch','grads_finalize','reduce_partials')])
PY
grep -E "adamw|convert|finalize|reduce_part|step_begin" gpurun_out/r3j/bench.txt gpurun_out/r3j/bench_unfused.txt
