cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
cd /tmp && export TMPDIR=/tmp
for m in 1 2; do
export AMT_BIG=$m
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3f/pmcA$m -o p -- $GRAFT_REPO_ROOT/scratch/big_bench_nt 112000 0 > $GRAFT_REPO_ROOT/gpurun_out/r3f/pmcA$m.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3f/pmcB$m -o p -- $GRAFT_REPO_ROOT/scratch/big_bench_nt 112000 0 > $GRAFT_REPO_ROOT/gpurun_out/r3f/pmcB$m.log 2>&1; echo rc=$?
done
cd $GRAFT_REPO_ROOT
find gpurun_out/r3f -name "*.db" -delete
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/r3f/pmc*')):
    if not d.endswith(('1','2')): continue
    fs = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no csv'); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'ffn_fwd_big' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    print(d, {k: round(sum(v[1:]) / max(1, len(v) - 1)) for k, v in acc.items()})
PY
