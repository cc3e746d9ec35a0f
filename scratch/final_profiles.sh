cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/r02_prof gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/r02_prof_bench.json 2> gpurun_out/r02_prof_bench.err; echo rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2> gpurun_out/r02_pmc_fetch.err; echo rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r02_pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2> gpurun_out/r02_pmc_write.err; echo rc=$?
python3 profiles/make_traffic.py gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write gpurun_out/r02_pmc_traffic_by_class.json > /dev/null
rm -rf gpurun_out/r02_pmc_fetch gpurun_out/r02_pmc_write
find gpurun_out/r02_prof -name "*kernel_trace.csv" -delete
cp gpurun_out/r02_prof/*/*_kernel_stats.csv gpurun_out/r02_kernel_stats_graph.csv
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench_breakdown.txt
tail -1 gpurun_out/r02_bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','loss','grads_valid')}, d['roofline'], d['cpu_baseline'], d['rooflines'])"
timeout -k 10 200 python bench.py --workload infer --steps 5 --warmup 2 > gpurun_out/r02_bench_infer.json 2> gpurun_out/r02_bench_infer.err
tail -1 gpurun_out/r02_bench_infer.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('infer', d['value'], d['ms_per_step'], d['rooflines'])"
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --clips-per-gpu 32 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('32 clips:', d['value'], d['ms_per_step'])"
