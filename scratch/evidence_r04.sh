# Round-4 evidence: tests, benches, rocprofv3 kernel stats (graph replay), PMC traffic, SQ (MFMA) counters.  On the GPU box:
#   gpurun --timeout 1190 -- 'bash scratch/evidence_r04.sh'      (part A: tests, bench lines, rocprofv3 kernel stats)
#   gpurun --timeout 1190 -- 'bash scratch/evidence_r04_pmc.sh'  (part B: PMC passes, final bench line with them, N > 1 rehearsal)
set -u
cd $GRAFT_REPO_ROOT
O=gpurun_out/ev4
rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests -q -m gpu > $O/gpu_tests.log 2>&1; echo tests rc=$?
tail -2 $O/gpu_tests.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench_breakdown.txt; echo bench rc=$?
timeout -k 10 300 python bench.py --resident-inputs --no-cpu-baseline > $O/bench_resident.json 2> /dev/null; echo resident rc=$?
timeout -k 10 300 python bench.py --workload infer > $O/bench_infer.json 2> $O/bench_infer_breakdown.txt; echo infer rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --clips-per-gpu 32 > $O/bench_b32.json 2> $O/bench_b32_breakdown.txt; echo b32 rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --force-overlap > $O/bench_overlap.json 2> /dev/null; echo overlap rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --force-overlap --no-signal > $O/bench_overlap_two_calls.json 2> /dev/null; echo overlap2 rc=$?
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fused-optimizer --force-overlap > $O/bench_overlap_round3_protocol.json 2> /dev/null; echo legacy rc=$?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats -o ks -- python3 $R/bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 0 > $R/$O/kstats.log 2>&1; echo kstats rc=$?
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/kstats_b32 -o ks -- python3 $R/bench.py --no-cpu-baseline --clips-per-gpu 32 --steps 10 --warmup 3 --profile-steps 0 > $R/$O/kstats_b32.log 2>&1; echo kstats_b32 rc=$?
cd $R
cp $O/kstats/ks_kernel_stats.csv $O/r04_kernel_stats_graph.csv 2>/dev/null || cp $O/kstats/*/*kernel_stats.csv $O/r04_kernel_stats_graph.csv
cp $O/kstats_b32/ks_kernel_stats.csv $O/r04_kernel_stats_b32.csv 2>/dev/null || cp $O/kstats_b32/*/*kernel_stats.csv $O/r04_kernel_stats_b32.csv
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
ls $O
python - <<'PY'
import json
for f in ('bench','bench_resident','bench_infer','bench_b32','bench_overlap','bench_overlap_two_calls','bench_overlap_round3_protocol'):
    try:
        d=json.loads(open(f'gpurun_out/ev4/{f}.json').read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_resident_inputs'), d.get('cpu_baseline') and d['cpu_baseline']['value'])
    except Exception as e: print(f, 'ERR', e)
PY
