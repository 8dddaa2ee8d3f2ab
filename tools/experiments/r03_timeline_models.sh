# One steady-state step, kernel by kernel, of the GAT and GraphSAGE bench requests (rocprofv3 kernel trace).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
for g in gat sage; do
  timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt_$g -o p -- python bench.py --gnn $g --steps 40 --warmup 10 --repeats 1 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/kt_$g.log 2>&1
  python tools/rocpd_timeline.py /tmp/pmc/kt_$g/p_results.db step_tail 10 > gpurun_out/r03_${g}_step_timeline.md 2>&1 || python tools/rocpd_timeline.py /tmp/pmc/kt_$g/p_results.db loss_finalize 10 > gpurun_out/r03_${g}_step_timeline.md 2>&1
  head -40 gpurun_out/r03_${g}_step_timeline.md | cut -c1-150
done
