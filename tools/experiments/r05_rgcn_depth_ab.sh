# wave-private R-GCN kernel: units of rows in flight per wave, now that the 128-float launches run with the XCD-block mapping
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out; rm -f gpurun_out/r05_rgcn_depth_ab.txt
for rep in 1 2; do
for d in default 1 3; do
  echo "GD_RGCN_WAVE_DEPTH=$d" >> gpurun_out/r05_rgcn_depth_ab.txt
  if [ $d = default ]; then unset GD_RGCN_WAVE_DEPTH; else export GD_RGCN_WAVE_DEPTH=$d; fi
  python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline --no_cached_rate --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r05_rgcn_depth_ab.txt
done; done
cat gpurun_out/r05_rgcn_depth_ab.txt
