# round 6, second GPU session: the new code paths' tests + the node-deletion bench line + the driver-shape bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "weight_stationary or gram or rbf or loss_zoo or spmm or gat_hub" --durations=8 2>&1 | tail -15
  timeout 1200 python -m pytest tests/test_engine_gpu.py -x -q --durations=8 2>&1 | tail -15
  timeout 1500 python -m pytest tests/test_full_size_gpu.py -x -q -s -k "collab-gcn or collab-gat or rgcn_forward or rgcn_fused" --durations=8 2>&1 | grep -v "^$" | tail -30
  timeout 900 python -m pytest tests/test_bench_gpu.py -x -q -k "node_deletion or launches_its_own" --durations=5 2>&1 | tail -8
  timeout 900 python -m pytest tests/test_long_parity_gpu.py -x -q -s -k "small-gcn" --durations=3 2>&1 | tail -12
  timeout 900 python -m pytest "tests/test_dist_cpu.py::test_partitioned_engine_matches_single_gpu_engine[3]" -x -q --durations=3 2>&1 | tail -5
) > gpurun_out/r06_b_tests.log 2>&1
tail -120 gpurun_out/r06_b_tests.log
timeout 900 python bench.py --workload synth-collab-nodecls --gnn gat --steps 100 --warmup 20 --cpu_baseline_iters 4 > gpurun_out/r06_b_nodecls_gat.json 2> gpurun_out/r06_b_nodecls_gat.err; tail -3 gpurun_out/r06_b_nodecls_gat.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r06_b_nodecls_gat.json') if l.startswith('{')][0])
    print('nodecls gat:', round(d['ms_per_step'], 4), 'ms', round(d['value'], 1), 'it/s; extras', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in d['extras'].items() if k != 'stage_rooflines'},
          'cpu', d.get('cpu_baseline', {}).get('value'), 'parity', d.get('parity'), 'cfg', {k: d['config'][k] for k in ('S1', 'S2', 'spmm_nnz', 'out_dim_padded_to', 'chained_del1', 'fused_layer2')})
except Exception as e:
    print('nodecls line failed', e)
PY
timeout 900 python bench.py --steps 20 --warmup 5 --no_cpu_baseline --pretrain_epochs 0 > gpurun_out/r06_b_driver_shape.json 2> gpurun_out/r06_b_driver_shape.err; tail -3 gpurun_out/r06_b_driver_shape.err
python - <<'PY'
import json
try:
    d = json.loads([l for l in open('gpurun_out/r06_b_driver_shape.json') if l.startswith('{')][0])
    ex = d['extras']
    print('driver shape:', round(d['ms_per_step'], 4), 'ms', round(d['value'], 1), 'it/s;', {k: round(v, 1) for k, v in ex.items() if k.startswith('iters_per_s')})
    print('regions', ex.get('ms_per_step_each_region'))
except Exception as e:
    print('driver-shape line failed', e)
PY
