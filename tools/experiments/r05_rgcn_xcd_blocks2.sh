# typed conv XCD-block mapping, per width (default: on for 128-float sources): step time + per-kernel durations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
OUT=gpurun_out/r05_rgcn_xcd_blocks2.txt
rm -f $OUT
ARGS="bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline --no_cached_rate"
for rep in 1 2; do
for mode in 0 default; do
  echo "GD_RGCN_WAVE_XCD_BLOCKS=$mode" >> $OUT
  if [ $mode = default ]; then unset GD_RGCN_WAVE_XCD_BLOCKS; else export GD_RGCN_WAVE_XCD_BLOCKS=$mode; fi
  python $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'final loss', d['final_loss'])" >> $OUT
done; done
unset GD_RGCN_WAVE_XCD_BLOCKS
rm -rf /tmp/pmc/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python $ARGS --steps 20 --warmup 4 --repeats 1 > /tmp/pmc/kt.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db loss_finalize 6 > gpurun_out/r05_rgcn_step_timeline.md 2>&1 || python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db step_tail 6 > gpurun_out/r05_rgcn_step_timeline.md 2>&1
cat $OUT; cat gpurun_out/r05_rgcn_step_timeline.md | head -30
