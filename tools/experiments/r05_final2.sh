# second half of r05_final.sh after the loss-partial buffer fix: the GCN profile, the bench lines that read it, the GAT line, the table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python bench.py --gnn gat --no_cpu_baseline --no_cached_rate --steps 5 --warmup 2 --pretrain_epochs 3 2>&1 | tail -1 | cut -c1-200
TAG=r05_final bash tools/experiments/r05_profile.sh > gpurun_out/r05_final_profile.log 2>&1
cp gpurun_out/r05_final_stages.json profiles/r05_final_stages.json
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python bench.py --steps 20 --warmup 5 --no_cached_rate > gpurun_out/r05_bench_driver_shape.json 2>> gpurun_out/r05_bench_default.err
python bench.py --gnn gat --no_cpu_baseline --no_cached_rate > gpurun_out/r05_bench_gat.json 2>> gpurun_out/r05_bench_default.err
bash tools/experiments/bench_table.sh > gpurun_out/r05_bench_table.txt 2>&1
tail -c 600 gpurun_out/r05_final_profile.log; cat gpurun_out/r05_bench_table.txt
python - <<'PY'
import json
for f in ('r05_bench_default', 'r05_bench_driver_shape', 'r05_bench_gat'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][0])
        print(f, round(d['value'], 1), round(d['ms_per_step'], 4), d['roofline'].get('frac'), d['roofline'].get('stage_profile'), (d.get('cpu_baseline') or {}).get('value'), d.get('speedup_vs_cpu'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
