cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0"
( echo "## separate reductions + finalize (GD_NO_STEP_TAIL=1)"; GD_NO_STEP_TAIL=1 $B | cut -c1-200
  echo "## step tail (default)"; $B | cut -c1-200
  echo "## separate again"; GD_NO_STEP_TAIL=1 $B | cut -c1-200
  echo "## step tail again"; $B | cut -c1-200 ) > gpurun_out/r03_tail_ab.txt 2>&1
cat gpurun_out/r03_tail_ab.txt
