cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
for sp in 6 0; do
export GD_MATRIX_SPLIT=$sp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt$sp -o p -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/kt$sp.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt$sp/p_results.db loss_finalize 25 > gpurun_out/r02_split${sp}_step_timeline.md 2>&1
done
