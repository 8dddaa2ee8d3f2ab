# the trainer's default step (layer-1 output cached + affected rows only) with the split / fused / chained Del-1 forms
# (GD_CACHE_SPLIT=1, default since round 6) against round 5's three-launch form (GD_CACHE_SPLIT=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q 2>&1 | tail -3
timeout 900 python -m pytest tests/test_cli_gpu.py -x -q -k "pipeline or node_deletion or reproduces" 2>&1 | tail -3
rm -f gpurun_out/r06_cache_split_ab.txt
for rep in 1 2; do
for v in 0 1; do
  echo "GD_CACHE_SPLIT=$v" >> gpurun_out/r06_cache_split_ab.txt
  GD_CACHE_SPLIT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); ex=d['extras']; print('gcn', round(d['value'],1), {k: round(v,1) for k,v in ex.items() if k.startswith('iters_per_s') and isinstance(v,float)})" >> gpurun_out/r06_cache_split_ab.txt
  GD_CACHE_SPLIT=$v timeout 600 python bench.py --workload synth-collab-nodecls --gnn gat --df_size 5 --steps 20 --warmup 5 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); ex=d['extras']; print('nodecls gat', round(d['value'],1), {k: round(v,1) for k,v in ex.items() if k.startswith('iters_per_s') and isinstance(v,float)})" >> gpurun_out/r06_cache_split_ab.txt
done; done
cat gpurun_out/r06_cache_split_ab.txt
