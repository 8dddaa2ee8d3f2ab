cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "32 32 8192" "40 24 8192" "40 24 2048" "48 16 8192"; do
  set -- $cfg
  GD_SPMM_RUN_WIN=$1 GD_SPMM_RUN_LIGHT=$2 GD_SPMM_RUNS_GRID=$3 timeout 600 python tools/experiments/spmm_runs_micro.py 2>&1 | grep -E "item kernel|Error|error" >> gpurun_out/r05_runs_micro2.txt
done
cat gpurun_out/r05_runs_micro2.txt
