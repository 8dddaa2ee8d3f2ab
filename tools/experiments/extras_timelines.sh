cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/tl2 -o p -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline --pretrain_epochs 0 > /tmp/pmc/tl2.log 2>&1
for b in 50 85 120; do python tools/rocpd_timeline.py /tmp/pmc/tl2/p_results.db loss_finalize $b > gpurun_out/r02_tl2_back$b.md 2>&1; done
