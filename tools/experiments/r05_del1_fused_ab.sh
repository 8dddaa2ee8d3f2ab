# Del-1 + layer-1 loss + W_D1 weight gradient in one pass (gd_del1_loss_wgrad_f32); GD_DEL1_FUSED=0: the two launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "del1_forward_loss" 2>&1 | tail -15 > gpurun_out/r05_del1_fused_test.log
cat gpurun_out/r05_del1_fused_test.log
rm -f gpurun_out/r05_del1_fused_ab.txt
for rep in 1 2 3; do
for mode in 0 1; do
  echo "GD_DEL1_FUSED=$mode" >> gpurun_out/r05_del1_fused_ab.txt
  GD_DEL1_FUSED=$mode timeout 600 python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'final loss', d['final_loss'])" >> gpurun_out/r05_del1_fused_ab.txt
done; done
cat gpurun_out/r05_del1_fused_ab.txt
rm -rf /tmp/pmc/kt
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --no_cpu_baseline --no_cached_rate --steps 20 --warmup 4 --repeats 1 > /tmp/pmc/kt.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db step_tail 6 > gpurun_out/r05_gcn_step_timeline.md 2>&1
head -20 gpurun_out/r05_gcn_step_timeline.md
