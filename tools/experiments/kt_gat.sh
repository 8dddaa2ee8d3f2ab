# kernel-trace timeline of one replayed step for another model (GNN=gat|gin|sage)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GNN=${GNN:-gat}
mkdir -p /tmp/pmc gpurun_out
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt_$GNN -o p -- python bench.py --gnn $GNN --steps 30 --warmup 5 --no_cpu_baseline --no_cached_rate > /tmp/pmc/kt_$GNN.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt_$GNN/p_results.db loss_finalize 10 > gpurun_out/timeline_$GNN.md 2>&1
grep metric /tmp/pmc/kt_$GNN.log | cut -c1-120 >> gpurun_out/timeline_$GNN.md
