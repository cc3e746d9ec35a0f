cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export AMT_CNN_WGRAD_LATE=1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_late -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --profile-steps 0 > /dev/null 2> gpurun_out/prof_late.err
python3 - <<PY
import csv,glob
f=glob.glob('gpurun_out/prof_late/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    if 'block_bwd' in n or 'gemm_tn' in n or 'small' in n: print(int(r['Calls']), round(float(r['AverageNs'])/1e3,2), n[:100])
PY
rm -rf gpurun_out/prof_late
