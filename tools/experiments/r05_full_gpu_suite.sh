cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r05_full_gpu_suite.log
tail -25 gpurun_out/r05_full_gpu_suite.log
