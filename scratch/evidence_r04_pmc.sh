# Round-4 evidence, part B (after scratch/evidence_r04.sh; needs profiles/r04_kernel_stats_graph.csv of THIS build in place):
# PMC traffic + SQ counters at cfg2, PMC traffic at 32 clips, the bench line that quotes them, the N > 1 rehearsal.
set -u
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/ev4b
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="--steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 --resident-inputs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py $P > $R/$O/pmc_fetch.log 2>&1; echo pmc_fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py $P > $R/$O/pmc_write.log 2>&1; echo pmc_write rc=$?
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$O/pmc_sq -- python3 $R/bench.py $P > $R/$O/pmc_sq.log 2>&1; echo pmc_sq rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch_b32 -- python3 $R/bench.py $P --clips-per-gpu 32 > $R/$O/pmc_fetch_b32.log 2>&1; echo pmc_fetch_b32 rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write_b32 -- python3 $R/bench.py $P --clips-per-gpu 32 > $R/$O/pmc_write_b32.log 2>&1; echo pmc_write_b32 rc=$?
cd $R
python profiles/make_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_by_class.json profiles/r04_kernel_stats_graph.csv > $O/make_traffic.log 2>&1; echo traffic rc=$?
python profiles/make_traffic.py $O/pmc_fetch_b32 $O/pmc_write_b32 $O/pmc_traffic_by_class_b32.json profiles/r04_kernel_stats_b32.csv > $O/make_traffic_b32.log 2>&1; echo traffic_b32 rc=$?
python profiles/make_mfma.py $O/pmc_sq $O/pmc_mfma_by_class.json > $O/make_mfma.log 2>&1; echo mfma rc=$?
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
# the bench line that quotes the counters of this build (as the driver's run will: they are committed under profiles/)
cp $O/pmc_traffic_by_class.json profiles/pmc_traffic_by_class.json; cp $O/pmc_mfma_by_class.json profiles/pmc_mfma_by_class.json
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench_breakdown.txt; echo bench rc=$?
for N in 2 4; do
  AMT_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600+N)) \
    bench.py --gpus $N --steps 10 --warmup 3 > $O/bench_rehearsal_n$N.json 2> $O/bench_rehearsal_n$N.err; echo rehearsal $N rc=$?
done
ls $O
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ev4b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_resident_inputs'), d['roofline'], d['cpu_baseline']['value'])
for f in ('pmc_traffic_by_class','pmc_traffic_by_class_b32'):
    j=json.load(open(f'gpurun_out/ev4b/{f}.json'))
    print(f, {k:v for k,v in j.items() if k.startswith('_')})
PY
# (round 3's protocol once more: the run in part A is sometimes hit by a one-off start-up hiccup of the world-1 RCCL group)
timeout -k 10 300 python bench.py --no-cpu-baseline --no-fused-optimizer --force-overlap > $O/bench_overlap_round3_protocol.json 2> /dev/null; echo legacy rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ev4b/bench_overlap_round3_protocol.json').read().strip().splitlines()[-1]); print('legacy', d['ms_per_step'], d.get('ms_per_step_resident_inputs'))
PY
