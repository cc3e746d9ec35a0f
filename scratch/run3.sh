cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_parity_full_gpu.py -q -m gpu -k "large_n or bf16_events or streaming" -s > gpurun_out/r3c/tests.log 2>&1; echo tests rc=$?
grep -E "passed|failed|bf16 events|Error|error" gpurun_out/r3c/tests.log | head -20
timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3c/infer.json 2> gpurun_out/r3c/infer.txt; echo infer rc=$?
grep -A14 "kernel class breakdown" gpurun_out/r3c/infer.txt
python -c "import json;d=json.loads(open('gpurun_out/r3c/infer.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_infer -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --workload infer --steps 5 --warmup 2 --profile-steps 0 > $GRAFT_REPO_ROOT/gpurun_out/r3c/prof_infer.log 2>&1; echo prof rc=$?
cd $GRAFT_REPO_ROOT
find gpurun_out/r3c/prof_infer -name "*kernel_trace.csv" -delete; find gpurun_out/r3c/prof_infer -name "*.db" -delete
cat gpurun_out/r3c/prof_infer/*/*kernel_stats.csv | cut -c1-200 | head -40
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline > gpurun_out/r3c/b32.json 2> gpurun_out/r3c/b32.txt; echo b32 rc=$?
python -c "import json;d=json.loads(open('gpurun_out/r3c/b32.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
