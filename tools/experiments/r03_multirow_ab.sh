cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
B="python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0"
( echo "## one row per item (GD_SPMM_MULTIROW=0)"; GD_SPMM_MULTIROW=0 $B | cut -c1-170
  echo "## multi-row items (default)"; $B | cut -c1-170
  echo "## one row per item again"; GD_SPMM_MULTIROW=0 $B | cut -c1-170
  echo "## multi-row items again"; $B | cut -c1-170 ) > gpurun_out/r03_multirow_ab.txt 2>&1
cat gpurun_out/r03_multirow_ab.txt
