cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3i
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3i/tests.log 2>&1; echo tests rc=$?
tail -3 gpurun_out/r3i/tests.log
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py $BARGS > gpurun_out/r3i/$name.json 2> gpurun_out/r3i/$name.txt; echo $name rc=$?; }
BARGS="--steps 20 --warmup 5 --no-cpu-baseline"; run bench A=1; run bench_nobits AMT_ATTN_BITS=0
BARGS="--workload infer --steps 5 --warmup 2"; run infer A=1; run infer_c256short AMT_BLOCK_INF_LONG=0; run infer_qkv32 AMT_QKV_FWD_TT=32
BARGS="--steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline"; run b32 A=1; run b32_ffnbwd16 AMT_FFN_BWD_TT=16
python - <<'PY'
import json,glob
for f in ('bench','bench_nobits','infer','infer_c256short','infer_qkv32','b32','b32_ffnbwd16'):
    d=json.loads(open(f'gpurun_out/r3i/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], [(b['kernel'], b['avg_us']) for b in d['breakdown'][:10]])
PY
