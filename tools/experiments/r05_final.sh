# Everything DESIGN.md quotes for round 5, from one box: the profiled default line (stage table with PMC traffic) for GCN, the
# in-step stage tables of GAT and GraphSAGE, the default bench line (with cpu_baseline and the parity leg), the driver-shaped
# line, the R-GCN line, the table of the other configurations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r05_final bash tools/experiments/r05_profile.sh > gpurun_out/r05_final_profile.log 2>&1
cp gpurun_out/r05_final_stages.json profiles/r05_final_stages.json        # (bench.py reads it from profiles/ for in_step_us / traffic)
for g in gat sage; do
  GNN=$g TAG=r05_final_$g bash tools/experiments/r05_profile.sh > gpurun_out/r05_final_${g}_profile.log 2>&1
  cp gpurun_out/r05_final_${g}_stages.json profiles/r05_final_stages_$g.json
done
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
python bench.py --steps 20 --warmup 5 --no_cached_rate > gpurun_out/r05_bench_driver_shape.json 2>> gpurun_out/r05_bench_default.err
python bench.py --gnn gat --no_cpu_baseline --no_cached_rate > gpurun_out/r05_bench_gat.json 2>> gpurun_out/r05_bench_default.err
python bench.py --gnn sage --no_cpu_baseline --no_cached_rate > gpurun_out/r05_bench_sage.json 2>> gpurun_out/r05_bench_default.err
python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 > gpurun_out/r05_bench_synth_biokg_rgcn.json 2>> gpurun_out/r05_bench_default.err
bash tools/experiments/bench_table.sh > gpurun_out/r05_bench_table.txt 2>&1
# one R-GCN step as the profiler sees it (launch list + per-kernel totals)
rm -rf /tmp/pmc/kt_rgcn
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt_rgcn -o p -- python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline --no_cached_rate --steps 20 --warmup 4 --repeats 1 > /tmp/pmc/kt_rgcn.log 2>&1
python tools/rocpd_timeline.py /tmp/pmc/kt_rgcn/p_results.db step_tail 6 > gpurun_out/r05_rgcn_step_timeline.md 2>&1
python tools/rocpd_summary.py /tmp/pmc/kt_rgcn/p_results.db gpurun_out/r05_rgcn_kernel_stats.md > /dev/null 2>&1
tail -c 400 gpurun_out/r05_final_profile.log; tail -c 300 gpurun_out/r05_final_gat_profile.log; tail -c 300 gpurun_out/r05_final_sage_profile.log; cat gpurun_out/r05_bench_table.txt
python - <<'PY'
import json
for f in ('r05_bench_default', 'r05_bench_driver_shape', 'r05_bench_gat', 'r05_bench_sage', 'r05_bench_synth_biokg_rgcn'):
    try:
        d = json.loads([l for l in open(f'gpurun_out/{f}.json') if l.startswith('{')][0])
        print(f, round(d['value'], 1), round(d['ms_per_step'], 4), d['roofline'].get('frac'), d['roofline'].get('stage_profile'), (d.get('cpu_baseline') or {}).get('value'), d.get('speedup_vs_cpu'))
    except Exception as e:
        print(f, 'FAILED', e)
PY
