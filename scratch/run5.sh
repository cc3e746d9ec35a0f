cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3e
for m in 1 2; do
AMT_BIG=$m timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3e/infer$m.json 2> gpurun_out/r3e/infer$m.txt; echo infer rc=$?
AMT_BIG=$m timeout -k 10 200 python bench.py --steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline > gpurun_out/r3e/b32_$m.json 2> gpurun_out/r3e/b32_$m.txt; echo b32 rc=$?
done
python - <<'PY'
import json
for f in ('infer1','infer2','b32_1','b32_2'):
    d=json.loads(open(f'gpurun_out/r3e/{f}.json').read().strip().splitlines()[-1])
    print(f, d['value'], d['ms_per_step'], [(b['kernel'], b['avg_us']) for b in d['breakdown'] if 'ffn' in b['kernel']])
PY
