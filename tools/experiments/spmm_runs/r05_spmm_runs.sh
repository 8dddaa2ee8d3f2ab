# run-item SpMM: parity test, then back-to-back timings of the step's three aggregations for a few windows / grids
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "run_item or onepass or rowgroup" 2>&1 | tail -15 > gpurun_out/r05_runs_test.log
for cfg in "32 32 8192" "16 16 8192" "48 16 8192" "24 40 8192" "32 32 4096" "32 32 2048" "32 32 1024" "40 24 8192"; do
  set -- $cfg
  GD_SPMM_RUN_WIN=$1 GD_SPMM_RUN_LIGHT=$2 GD_SPMM_RUNS_GRID=$3 timeout 600 python tools/experiments/spmm_runs_micro.py 2>&1 | grep -E "item kernel|Error|error" >> gpurun_out/r05_runs_micro.txt
done
cat gpurun_out/r05_runs_test.log; cat gpurun_out/r05_runs_micro.txt
