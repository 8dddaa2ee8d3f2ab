# A/B of the round-3 step changes on one box: hub rows inside the SpMM launch, tile queue in the row GEMMs.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
B="python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0"
( echo "## two-launch SpMM, static tiles"; GD_SPMM_TWO_LAUNCH=1 GD_ROWS_GEMM_QUEUE=0 $B | cut -c1-220
  echo "## onepass SpMM, static tiles"; GD_ROWS_GEMM_QUEUE=0 $B | cut -c1-220
  echo "## two-launch SpMM, tile queue"; GD_SPMM_TWO_LAUNCH=1 $B | cut -c1-220
  echo "## both on (default)"; $B | cut -c1-220
  echo "## default, 20 steps x3"; for i in 1 2 3; do python bench.py --steps 20 --warmup 5 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 | cut -c1-160; done
) > gpurun_out/r03_ab.txt 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/kt.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db loss_finalize 25 > gpurun_out/r03_a_step_timeline.md 2>&1
python tools/rocpd_summary.py /tmp/pmc/kt/p_results.db gpurun_out/r03_a_kernel_stats.md > /dev/null
