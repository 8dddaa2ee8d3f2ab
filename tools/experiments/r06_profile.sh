# The bench command under rocprofv3: kernel trace (+ per-kernel stats, one-step timeline) and two separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE) as the MI355X guide prescribes; tools/rocpd_stage_table.py folds them into the per-stage table
# bench.py attaches to its roofline entries.  The table is stamped with the hash of the kernel sources it measured
# (bench.kernel_source_hash): bench.py refuses a table of other kernels.
#   TAG  names the outputs (gpurun_out/${TAG}_*; copy to profiles/)         GNN  gcn (default) | gat | sage | gin
#   WORKLOAD  synth-collab (default) | synth-collab-nodecls | synth-dblp ...   EXTRA  further bench.py arguments (e.g. "--df out --df_size 2.5")
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
TAG=${TAG:-r06_final}
GNN=${GNN:-gcn}
WORKLOAD=${WORKLOAD:-synth-collab}
EXTRA=${EXTRA:-}
STAGES=${STAGES:-$([ "$GNN" = gcn ] && [ "$WORKLOAD" = synth-collab ] && echo xw1,spmm1,del1_loss_wgrad1,t2,spmm2,del2_loss_bwd,spmm2_t,tail || echo auto)}
ARGS="bench.py --workload $WORKLOAD $EXTRA --gnn $GNN --steps 40 --warmup 10 --repeats 1 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0"
rm -rf /tmp/pmc/kt /tmp/pmc/f /tmp/pmc/w
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python $ARGS > /tmp/pmc/kt.log 2>&1
python tools/rocpd_summary.py /tmp/pmc/kt/p_results.db gpurun_out/${TAG}_kernel_stats.md > /dev/null
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db step_tail 10 > gpurun_out/${TAG}_step_timeline.md 2>&1
grep metric /tmp/pmc/kt.log > gpurun_out/${TAG}_bench_under_rocprof.json
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc/f -o p -- python $ARGS > /tmp/pmc/f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc/w -o p -- python $ARGS > /tmp/pmc/w.log 2>&1
python tools/rocpd_stage_table.py /tmp/pmc/kt/p_results.db --fetch /tmp/pmc/f/p_results.db --write /tmp/pmc/w/p_results.db --stages $STAGES --out /tmp/pmc/stages.json > /dev/null 2> /tmp/pmc/stages.err || cat /tmp/pmc/stages.err
TAG=$TAG GNN=$GNN WORKLOAD=$WORKLOAD EXTRA="$EXTRA" python - <<'PY'
import json, os, sys
sys.path.insert(0, '.')
import bench
tag, gnn = os.environ['TAG'], os.environ['GNN']
st = json.load(open('/tmp/pmc/stages.json'))
line = json.loads(open(f'gpurun_out/{tag}_bench_under_rocprof.json').read().strip().splitlines()[-1])
cfg = line['config']
st['workload'] = {'num_nodes': cfg['num_nodes'], 'spmm_nnz': cfg['spmm_nnz'], 'S1': cfg['S1'], 'S2': cfg['S2'], 'what': cfg['workload'], 'gnn': gnn}
st['csrc_sha'] = bench.kernel_source_hash()
wl, extra = os.environ['WORKLOAD'], os.environ.get('EXTRA', '')
st['command'] = (f'rocprofv3 --kernel-trace --stats -- python bench.py --workload {wl} {extra} --gnn {gnn} --steps 40 --warmup 10 --repeats 1 --no_cpu_baseline --no_cached_rate '
                 '--pretrain_epochs 0 ; the same under --pmc FETCH_SIZE and under --pmc WRITE_SIZE (separate passes; tools/experiments/r06_profile.sh)')
st['correction'] = 'gfx950: traffic_bytes = 2 x FETCH_SIZE KiB (32-B requests counted where 64 B move) + WRITE_SIZE KiB, x 1024'
json.dump(st, open(f'gpurun_out/{tag}_stages.json', 'w'), indent=1)
print(json.dumps({k: (round(v['in_step_us'], 1), v.get('traffic_bytes')) for k, v in st['stages'].items()}))
print('sum', st['sum_in_step_us'], 'span', st['step_span_us'], 'under rocprof', line['ms_per_step'])
PY
