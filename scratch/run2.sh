cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r3b/gpu_tests.log 2>&1; echo tests rc=$?
tail -15 gpurun_out/r3b/gpu_tests.log
