# One steady-state step, kernel by kernel, for a (workload, gnn) pair:  WL=synth-collab GNN=sage bash tools/experiments/step_timeline.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
WL=${WL:-synth-collab}; GNN=${GNN:-gcn}
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/tl_${WL}_${GNN} -o p -- python bench.py --workload $WL --gnn $GNN --steps 30 --warmup 5 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/tl.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/tl_${WL}_${GNN}/p_results.db loss_finalize 25 > gpurun_out/${TAG:-r02}_${WL}_${GNN}_step_timeline.md 2>&1
