# Round-3 evidence: tests, benches, rocprofv3 kernel stats (graph replay), PMC traffic.  Run on the GPU box:
#   gpurun --timeout 1190 -- 'bash scratch/evidence_r03.sh'
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/ev3
rm -rf $O; mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo tests rc=$?
tail -2 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench_breakdown.txt; echo bench rc=$?
timeout -k 10 300 python bench.py --workload infer > $O/bench_infer.json 2> $O/bench_infer_breakdown.txt; echo infer rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --clips-per-gpu 32 > $O/bench_b32.json 2> $O/bench_b32_breakdown.txt; echo b32 rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fused-optimizer > $O/bench_unfused_opt.json 2> /dev/null; echo unfused rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 0 > $GRAFT_REPO_ROOT/$O/kstats.log 2>&1; echo kstats rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats_infer -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --workload infer --steps 5 --warmup 2 --profile-steps 0 > $GRAFT_REPO_ROOT/$O/kstats_infer.log 2>&1; echo kstats_infer rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1; echo pmc_fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1; echo pmc_write rc=$?
cd $GRAFT_REPO_ROOT
cp $O/kstats/ks_kernel_stats.csv $O/r03_kernel_stats_graph.csv 2>/dev/null || cp $O/kstats/*/*kernel_stats.csv $O/r03_kernel_stats_graph.csv
cp $O/kstats_infer/ks_kernel_stats.csv $O/r03_kernel_stats_infer.csv 2>/dev/null || cp $O/kstats_infer/*/*kernel_stats.csv $O/r03_kernel_stats_infer.csv
python profiles/make_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_by_class.json $O/r03_kernel_stats_graph.csv > $O/make_traffic.log 2>&1; echo traffic rc=$?
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
ls $O
python - <<'PY'
import json
for f in ('bench','bench_infer','bench_b32','bench_unfused_opt'):
    try:
        d=json.loads(open(f'gpurun_out/ev3/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('roofline') and {k:d['roofline'][k] for k in ('kernel','bound','achieved','frac','traffic','avg_launch_us')}, d.get('cpu_baseline') and d['cpu_baseline']['value'])
    except Exception as e: print(f, 'ERR', e)
PY
