cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/dblp -o p -- python bench.py --workload synth-dblp --steps 30 --warmup 5 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/dblp.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/dblp/p_results.db loss_finalize 25 > gpurun_out/r02_dblp_step_timeline.md 2>&1
tail -2 /tmp/pmc/dblp.log | cut -c1-200
