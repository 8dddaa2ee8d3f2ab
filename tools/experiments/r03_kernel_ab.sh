# Same box, same bench command under rocprofv3 with and without an environment switch: per-kernel averages side by side.
#   VAR="GD_SPMM_TWO_LAUNCH=1" bash tools/experiments/r03_kernel_ab.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
ARGS="bench.py --steps 60 --warmup 10 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/ab_a -o p -- python $ARGS > /tmp/pmc/ab_a.log 2>&1
( export $VAR; timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/ab_b -o p -- python $ARGS > /tmp/pmc/ab_b.log 2>&1 )
python tools/rocpd_summary.py /tmp/pmc/ab_a/p_results.db /tmp/pmc/ab_a.md > /dev/null
python tools/rocpd_summary.py /tmp/pmc/ab_b/p_results.db /tmp/pmc/ab_b.md > /dev/null
( echo "# default"; grep -o '"value": [0-9.]*' /tmp/pmc/ab_a.log | head -1; grep "gd::" /tmp/pmc/ab_a.md | cut -c1-60,100-200 | head -16
  echo "# $VAR"; grep -o '"value": [0-9.]*' /tmp/pmc/ab_b.log | head -1; grep "gd::" /tmp/pmc/ab_b.md | cut -c1-60,100-200 | head -16 ) > gpurun_out/r03_kernel_ab_${TAG:-x}.txt 2>&1
cat gpurun_out/r03_kernel_ab_${TAG:-x}.txt
