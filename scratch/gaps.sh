cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/gaps
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --steps 6 --warmup 3 --no-cpu-baseline --profile-steps 0 --force-overlap > gpurun_out/gaps.json 2> gpurun_out/gaps.err
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/gaps/*/*kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# last ~2 steps
tail=rows[-520:]
gaps=[]
for (s0,e0,n0),(s1,e1,n1) in zip(tail,tail[1:]):
    g=s1-e0
    if g>4000: gaps.append((g,n0[:60],n1[:60]))
for g in gaps: print(round(g[0]/1e3,1),'us |',g[1],'->',g[2])
print('span of last 520 kernels (us):',(tail[-1][1]-tail[0][0])/1e3)
PY
rm -rf gpurun_out/gaps
tail -1 gpurun_out/gaps.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
