# the it/s table of DESIGN.md section 7: every model on synth-collab, GCN/GAT/GIN on the two small stand-ins
cd $GRAFT_REPO_ROOT
run() { python bench.py "$@" --no_cpu_baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(d['config']['workload'][:60], '|', round(d['value'], 1), 'it/s |', round(d['ms_per_step'], 3), 'ms | cached',
      round(d.get('extras', {}).get('iters_per_s_with_loop_invariant_layer1_cached', 0), 1))"; }
for g in gcn gat gin sage; do run --gnn $g; done
for g in gcn gat gin; do run --gnn $g --workload synth-dblp --df out --df_size 2.5; done
run --gnn gcn --workload synth-cora --df out --df_size 0.5
