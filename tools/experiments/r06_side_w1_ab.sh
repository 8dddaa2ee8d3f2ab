# small requests: the loss-fused W_D1 weight-gradient launch on a side stream next to layer 2's forward (GD_SMALL_SIDE_W1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -3
rm -f gpurun_out/r06_side_w1_ab.txt
for rep in 1 2 3; do
for v in 0 1; do
  for w in "synth-dblp --df out --df_size 2.5" "synth-cora --df out --df_size 0.5"; do
  echo "GD_SMALL_SIDE_W1=$v $w" >> gpurun_out/r06_side_w1_ab.txt
  GD_SMALL_SIDE_W1=$v timeout 600 python bench.py --workload $w --steps 400 --warmup 40 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r06_side_w1_ab.txt
  done
done; done
cat gpurun_out/r06_side_w1_ab.txt
