#!/bin/bash
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gaps
rm -rf $O; mkdir -p $O
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/res -- python3 bench.py --steps 30 --warmup 5 --resident-inputs --no-cpu-baseline --profile-steps 0 > $O/res.json 2> $O/res.err
python3 scratch/trace_gap_context.py $(ls $O/res/*/*kernel_trace.csv | head -1) > $O/res_ctx.txt
rm -rf $O/res
cat $O/res_ctx.txt
