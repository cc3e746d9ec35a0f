#!/bin/bash
# kernel trace of the default bench (fresh batch per step) and of the resident-input form: where is the device idle?
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/fresh -- python3 bench.py --steps 20 --warmup 5 > $O/fresh.json 2> $O/fresh.err
python3 scratch/trace_gaps.py $(ls $O/fresh/*/*kernel_trace.csv | head -1) > $O/fresh_gaps.txt
rocprofv3 --kernel-trace --output-format csv -d $O/res -- python3 bench.py --steps 20 --warmup 5 --resident-inputs > $O/res.json 2> $O/res.err
python3 scratch/trace_gaps.py $(ls $O/res/*/*kernel_trace.csv | head -1) > $O/res_gaps.txt
rm -rf $O/fresh $O/res
tail -22 $O/fresh_gaps.txt; tail -22 $O/res_gaps.txt
