cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/ks
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-steps 0 > gpurun_out/ks.json 2> gpurun_out/ks.err
find gpurun_out/ks -name "*kernel_trace.csv" -delete
cp gpurun_out/ks/*/*_kernel_stats.csv gpurun_out/ks_stats.csv
rm -rf gpurun_out/ks
tail -1 gpurun_out/ks.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
