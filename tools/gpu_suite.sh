# the whole -m gpu suite, log under gpurun_out/
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1100 python -m pytest tests -m gpu -q -x --timeout 900 > gpurun_out/r3_gpu_suite.log 2>&1
tail -15 gpurun_out/r3_gpu_suite.log
