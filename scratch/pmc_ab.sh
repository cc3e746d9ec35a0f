cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  export AMT_WGRAD_XCD=$v
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f$v -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > /dev/null 2> gpurun_out/pmc_f$v.err
  python3 - <<PY
import csv,glob,collections,sys
sys.path.insert(0,'profiles')
from make_traffic import classify
f=glob.glob('gpurun_out/pmc_f$v/*/*_counter_collection.csv')[0]
acc=collections.defaultdict(lambda:[0.0,0])
for r in csv.DictReader(open(f)):
    if r['Counter_Name']!='FETCH_SIZE': continue
    c=classify(r['Kernel_Name'])
    if c and 'wgrad' in c: acc[c][0]+=float(r['Counter_Value']); acc[c][1]+=1
for c,(v,n) in acc.items(): print('xcd=$v',c,'fetch MB/launch (x2 corrected):',round(2*v*1024/n/1e6,1),n)
PY
  rm -rf gpurun_out/pmc_f$v
done
