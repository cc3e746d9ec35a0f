# Round-5 evidence, part A: tests, bench lines, rocprofv3 kernel stats under hipGraph replay.  Every run keeps its stderr and
# its return code (no retries: ADVICE r4).   gpurun --timeout 1190 -- 'bash scratch/evidence_r05.sh'
set -u
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/ev5
rm -rf $O; mkdir -p $O
timeout -k 10 700 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo tests rc=$?
tail -2 $O/gpu_tests.log
run() { name=$1; shift; timeout -k 10 300 "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
run bench python bench.py
run bench_resident python bench.py --resident-inputs --no-cpu-baseline
run bench_labels_fp32 python bench.py --labels-fp32 --no-cpu-baseline
AMT_TLAYER=1 run bench_tlayer python bench.py --no-cpu-baseline
run bench_infer python bench.py --workload infer
run bench_b32 python bench.py --no-cpu-baseline --clips-per-gpu 32
run bench_overlap python bench.py --no-cpu-baseline --force-overlap
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -o ks -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 0 > $R/$O/kstats.log 2>&1; echo kstats rc=$?
cd $R
cp $O/kstats/ks_kernel_stats.csv $O/r05_kernel_stats_graph.csv 2>/dev/null || cp $O/kstats/*/*kernel_stats.csv $O/r05_kernel_stats_graph.csv
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
AMT_TLAYER=1 AMT_TLAYER_STAMPS=1 timeout -k 10 200 python scratch/tlayer_stamps.py > $O/r05_tlayer_stamps.txt 2> $O/tlayer_stamps.err; echo stamps rc=$?
ls $O
python - <<'PY'
import json
for f in ('bench','bench_resident','bench_labels_fp32','bench_tlayer','bench_infer','bench_b32','bench_overlap'):
    try:
        d=json.loads(open(f'gpurun_out/ev5/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_resident_inputs'), d.get('cpu_baseline') and d['cpu_baseline']['value'])
    except Exception as e: print(f, 'ERR', e)
PY
