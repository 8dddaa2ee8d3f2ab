# R-GCN step with the loss rows split into (inside the Del rows: fused forms) + (outside: one stand-alone loss launch per layer); GD_NO_SPLIT_LOSS=1: before
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_full_size_gpu.py -x -q -k "rgcn or kg" 2>&1 | tail -12 > gpurun_out/r05_rgcn_split_loss_test.log
rm -f gpurun_out/r05_rgcn_split_loss_ab.txt
for rep in 1 2 3; do
for mode in 1 0; do
  echo "GD_NO_SPLIT_LOSS=$mode" >> gpurun_out/r05_rgcn_split_loss_ab.txt
  GD_NO_SPLIT_LOSS=$mode python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline --no_cached_rate --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'final loss', d['final_loss'])" >> gpurun_out/r05_rgcn_split_loss_ab.txt
done; done
rm -rf /tmp/pmc/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline --no_cached_rate --steps 20 --warmup 4 --repeats 1 > /tmp/pmc/kt.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db step_tail 6 > gpurun_out/r05_rgcn_step_timeline.md 2>&1
cat gpurun_out/r05_rgcn_split_loss_test.log; cat gpurun_out/r05_rgcn_split_loss_ab.txt; head -28 gpurun_out/r05_rgcn_step_timeline.md
