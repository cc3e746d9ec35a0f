cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3a/gpu_tests.log 2>&1; echo tests rc=$?
tail -5 gpurun_out/r3a/gpu_tests.log
timeout -k 10 500 python scratch/parity_probe.py > gpurun_out/r3a/parity_probe.log 2>&1; echo probe rc=$?
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.txt; echo bench rc=$?
timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r3a/infer.json 2> gpurun_out/r3a/infer.txt; echo infer rc=$?
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --clips-per-gpu 32 --no-cpu-baseline > gpurun_out/r3a/b32.json 2> gpurun_out/r3a/b32.txt; echo b32 rc=$?
tail -3 gpurun_out/r3a/parity_probe.log
