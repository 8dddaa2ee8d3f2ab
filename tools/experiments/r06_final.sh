# Everything DESIGN.md quotes for round 6, from one box: the profiled default line (stage table with PMC traffic) for GCN, the
# in-step stage tables of GAT / GraphSAGE / the node-deletion GAT request (config 5) / synth-dblp (config 2), the default bench
# line (with cpu_baseline and the parity leg), the driver-shaped line, the node-deletion line, the R-GCN line, the table of the
# other configurations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r06_final bash tools/experiments/r06_profile.sh > gpurun_out/r06_final_profile.log 2>&1
cp gpurun_out/r06_final_stages.json profiles/r06_final_stages.json        # (bench.py reads it from profiles/ for in_step_us / traffic)
for g in gat sage; do
  GNN=$g TAG=r06_final_$g bash tools/experiments/r06_profile.sh > gpurun_out/r06_final_${g}_profile.log 2>&1
  cp gpurun_out/r06_final_${g}_stages.json profiles/r06_final_stages_$g.json
done
WORKLOAD=synth-collab-nodecls EXTRA="--df_size 5" GNN=gat TAG=r06_final_collab_nodecls_gat bash tools/experiments/r06_profile.sh > gpurun_out/r06_final_collab_nodecls_gat_profile.log 2>&1
cp gpurun_out/r06_final_collab_nodecls_gat_stages.json profiles/r06_final_stages_collab_nodecls_gat.json
WORKLOAD=synth-dblp EXTRA="--df out --df_size 2.5" TAG=r06_final_dblp bash tools/experiments/r06_profile.sh > gpurun_out/r06_final_dblp_profile.log 2>&1
cp gpurun_out/r06_final_dblp_stages.json gpurun_out/r06_final_stages_dblp.json
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_shape.json 2>> gpurun_out/r06_bench_default.err
python bench.py --gnn gat --no_cpu_baseline --no_cached_rate > gpurun_out/r06_bench_gat.json 2>> gpurun_out/r06_bench_default.err
python bench.py --gnn sage --no_cpu_baseline --no_cached_rate > gpurun_out/r06_bench_sage.json 2>> gpurun_out/r06_bench_default.err
python bench.py --workload synth-collab-nodecls --gnn gat --df_size 5 > gpurun_out/r06_bench_nodecls_gat.json 2>> gpurun_out/r06_bench_default.err
python bench.py --workload synth-collab-nodecls --gnn gat --df_size 5 --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r06_bench_nodecls_gat_driver_shape.json 2>> gpurun_out/r06_bench_default.err
python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 > gpurun_out/r06_bench_synth_biokg_rgcn.json 2>> gpurun_out/r06_bench_default.err
bash tools/experiments/bench_table.sh > gpurun_out/r06_bench_table.txt 2>&1
tail -c 400 gpurun_out/r06_final_profile.log; tail -c 300 gpurun_out/r06_final_gat_profile.log; tail -c 300 gpurun_out/r06_final_sage_profile.log
tail -c 300 gpurun_out/r06_final_collab_nodecls_gat_profile.log; tail -c 300 gpurun_out/r06_final_dblp_profile.log; cat gpurun_out/r06_bench_table.txt
python - <<'PY'
import json
for f in ('r06_bench_default', 'r06_bench_driver_shape', 'r06_bench_gat', 'r06_bench_sage', 'r06_bench_nodecls_gat', 'r06_bench_nodecls_gat_driver_shape', 'r06_bench_synth_biokg_rgcn'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][0])
        ex = d.get('extras', {})
        print(f, round(d['value'], 1), round(d['ms_per_step'], 4), d['roofline'].get('frac'), d['roofline'].get('stage_profile'), (d.get('cpu_baseline') or {}).get('value'), d.get('speedup_vs_cpu'),
              {k: round(v, 1) for k, v in ex.items() if k.startswith('iters_per_s') and isinstance(v, float)})
    except Exception as e:
        print(f, 'FAILED', e)
PY
