# Round-5 evidence, part B (after part A; needs profiles/r05_kernel_stats_graph.csv of THIS build in place): PMC traffic + SQ
# counters at cfg2, the FETCH / WRITE calibration, the bench line that quotes the counters, the N = 2 rehearsal.
#   gpurun --timeout 1190 -- 'bash scratch/evidence_r05_pmc.sh'
set -u
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
O=gpurun_out/ev5b
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
P="--steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 --resident-inputs"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_fetch -- python3 $R/bench.py $P > $R/$O/pmc_fetch.log 2>&1; echo pmc_fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/$O/pmc_write -- python3 $R/bench.py $P > $R/$O/pmc_write.log 2>&1; echo pmc_write rc=$?
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$O/pmc_sq -- python3 $R/bench.py $P > $R/$O/pmc_sq.log 2>&1; echo pmc_sq rc=$?
cd $R
python profiles/make_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_by_class.json profiles/r05_kernel_stats_graph.csv > $O/make_traffic.log 2>&1; echo traffic rc=$?
python profiles/make_mfma.py $O/pmc_sq $O/pmc_mfma_by_class.json > $O/make_mfma.log 2>&1; echo mfma rc=$?
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
bash scratch/fetch_calib.sh > $O/r05_fetch_calibration.txt 2>&1; echo calib rc=$?
cp $O/pmc_traffic_by_class.json profiles/pmc_traffic_by_class.json; cp $O/pmc_mfma_by_class.json profiles/pmc_mfma_by_class.json
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench_breakdown.txt; echo bench rc=$?
AMT_BENCH_REHEARSE_ON_ONE_GPU=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29602 \
    bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_rehearsal_n2.json 2> $O/bench_rehearsal_n2.err; echo rehearsal rc=$?
ls $O
python - <<'PY'
import json
d=json.loads(open('gpurun_out/ev5b/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('ms_per_step_resident_inputs'), d['roofline'], d['cpu_baseline']['value'])
j=json.load(open('gpurun_out/ev5b/pmc_traffic_by_class.json'))
print({k:v for k,v in j.items() if k.startswith('_')})
d=json.loads(open('gpurun_out/ev5b/bench_rehearsal_n2.json').read().strip().splitlines()[-1]); print('rehearsal', d['ms_per_step'], d['config'].get('grad_allreduce'), d['config'].get('grad_allreduce_fallback'))
PY
