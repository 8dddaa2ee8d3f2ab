cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 > gpurun_out/r06_smoke.log
timeout 2400 python -m pytest tests -x -q -m gpu --durations=25 2>&1 | tail -60 > gpurun_out/r06_full_gpu_suite.log
cat gpurun_out/r06_smoke.log; tail -45 gpurun_out/r06_full_gpu_suite.log
