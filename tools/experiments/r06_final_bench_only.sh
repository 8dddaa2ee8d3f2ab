# the bench lines of r06_final.sh again (engine-side change after the profiles were taken: the kernels - and the stage tables'
# source hash - are unchanged)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_shape.json 2>> gpurun_out/r06_bench_default.err
python bench.py --gnn gat --no_cpu_baseline > gpurun_out/r06_bench_gat.json 2>> gpurun_out/r06_bench_default.err
python bench.py --gnn sage --no_cpu_baseline > gpurun_out/r06_bench_sage.json 2>> gpurun_out/r06_bench_default.err
python bench.py --workload synth-collab-nodecls --gnn gat --df_size 5 > gpurun_out/r06_bench_nodecls_gat.json 2>> gpurun_out/r06_bench_default.err
python bench.py --workload synth-collab-nodecls --gnn gat --df_size 5 --steps 20 --warmup 5 --no_cpu_baseline > gpurun_out/r06_bench_nodecls_gat_driver_shape.json 2>> gpurun_out/r06_bench_default.err
python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 > gpurun_out/r06_bench_synth_biokg_rgcn.json 2>> gpurun_out/r06_bench_default.err
bash tools/experiments/bench_table.sh > gpurun_out/r06_bench_table.txt 2>&1
cat gpurun_out/r06_bench_table.txt
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
