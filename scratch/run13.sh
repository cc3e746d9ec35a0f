cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3m
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3m/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3m/tests.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3m/$name.json 2> gpurun_out/r3m/$name.txt; echo $name rc=$?; }
run infer A=1
run infer_ng2 AMT_ATTN_FWD_NG=2
run infer_c128short AMT_BLOCK_INF_LONG=5
run infer_c64short AMT_BLOCK_INF_LONG=6
python - <<'PY'
import json
for f in ('infer','infer_ng2','infer_c128short','infer_c64short'):
    d=json.loads(open(f'gpurun_out/r3m/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], [(b['kernel'], b['avg_us']) for b in d['breakdown'][:7]])
PY
