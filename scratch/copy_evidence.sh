# copies the merged gpurun_out/ev3 summaries into profiles/ (tracked)
set -e
cd /root/repo
O=gpurun_out/ev3
cp $O/bench.json profiles/r03_bench.json
cp $O/bench_breakdown.txt profiles/r03_bench_breakdown.txt
cp $O/bench_infer.json profiles/r03_bench_infer.json
cp $O/bench_infer_breakdown.txt profiles/r03_bench_infer_breakdown.txt
cp $O/bench_b32.json profiles/r03_bench_b32.json
cp $O/bench_unfused_opt.json profiles/r03_bench_unfused_optimizer.json
cp $O/r03_kernel_stats_graph.csv profiles/r03_kernel_stats_graph.csv
cp $O/r03_kernel_stats_infer.csv profiles/r03_kernel_stats_infer.csv
cp $O/pmc_traffic_by_class.json profiles/pmc_traffic_by_class.json
cp $O/pmc_traffic_by_class.json profiles/r03_pmc_traffic.json
tail -3 $O/gpu_tests.log > profiles/r03_gpu_tests_tail.txt
ls -la profiles | grep r03
