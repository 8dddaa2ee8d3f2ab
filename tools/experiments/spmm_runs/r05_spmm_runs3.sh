cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "run_item" 2>&1 | tail -15 > gpurun_out/r05_runs_test.log
rm -f gpurun_out/r05_runs_micro3.txt
for cfg in "32 32 8192" "40 24 8192" "40 24 2048" "40 24 1280"; do
  set -- $cfg
  GD_SPMM_RUN_WIN=$1 GD_SPMM_RUN_LIGHT=$2 GD_SPMM_RUNS_GRID=$3 timeout 600 python tools/experiments/spmm_runs_micro.py 2>&1 | grep -E "item kernel|Error|error" >> gpurun_out/r05_runs_micro3.txt
done
cat gpurun_out/r05_runs_test.log; cat gpurun_out/r05_runs_micro3.txt
