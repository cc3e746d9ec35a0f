#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ldspmc; rm -rf $O; mkdir -p $O
P="--steps 3 --warmup 1 --no-cpu-baseline --no-graph --profile-steps 0 --resident-inputs"
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/b32 -- python3 $R/bench.py $P --clips-per-gpu 32 > $O/b32.log 2>&1; echo rc=$?
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL --kernel-trace --output-format csv -d $O/b32v -- python3 $R/bench.py $P --clips-per-gpu 32 > $O/b32v.log 2>&1; echo rc=$?
cd $R
python profiles/make_counters.py $O/b32 $O/lds_b32.json > $O/lds_b32.txt 2>&1
python profiles/make_counters.py $O/b32v $O/vmem_b32.json > $O/vmem_b32.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
grep "gemm_tn\|block_bwd\|ffn_bwd\|ffn_fwd_big\|block_fwd" $O/lds_b32.txt $O/vmem_b32.txt
