set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/ev
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gpu_tests.log 2>&1
python bench.py > $O/bench.json 2> $O/bench_breakdown.txt
python bench.py --workload infer > $O/bench_infer.json 2> $O/bench_infer_breakdown.txt
python bench.py --no-cpu-baseline --attn-fp8 > $O/bench_fp8.json 2> $O/bench_fp8.txt
python bench.py --no-cpu-baseline --clips-per-gpu 32 > $O/bench_b32.json 2> $O/bench_b32.txt
python bench.py --no-cpu-baseline --clips-per-gpu 32 --attn-fp8 > $O/bench_b32_fp8.json 2> $O/bench_b32_fp8.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/kstats -o ks -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 --profile-steps 0 > $GRAFT_REPO_ROOT/$O/kstats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 > $GRAFT_REPO_ROOT/$O/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/make_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic_by_class.json > $O/make_traffic.log 2>&1
rm -rf $O/pmc_fetch/*/*.db $O/pmc_write/*/*.db
ls -la $O $O/kstats/* | head -40
tail -2 $O/gpu_tests.log
cut -c1-220 $O/bench.json
