# BASELINE config 4 end to end with the fused full-graph R-GCN step behind the CLI: prepare -> short original training ->
# delete_gnn.py --gnn rgcn --fullgraph (EP1 epochs, one validation at the end) -> test; wall times.
cd $GRAFT_REPO_ROOT
W=/tmp/kgrun; rm -rf $W; mkdir -p $W; cd $W
export PYTHONPATH=$GRAFT_REPO_ROOT
DS=${DS:-synth-biokg}
t0=$(date +%s.%N)
python $GRAFT_REPO_ROOT/prepare_dataset.py --dataset $DS --seeds 42 2>&1 | tail -1 | cut -c1-200
t1=$(date +%s.%N); echo "prepare: $(python -c "print(round($t1 - $t0, 1))") s"
GNNDELETE_FORCE_EPOCHS=2 GNNDELETE_FORCE_VALID_FREQ=2 timeout 1500 python $GRAFT_REPO_ROOT/train_gnn.py --dataset $DS --gnn rgcn --random_seed 42 2>&1 | tail -2 | cut -c1-300
t2=$(date +%s.%N); echo "train_gnn (2 epochs + eval + test): $(python -c "print(round($t2 - $t1, 1))") s"
GNNDELETE_FORCE_EPOCHS=${EP1:-100} GNNDELETE_FORCE_VALID_FREQ=${EP1:-100} timeout 1500 python $GRAFT_REPO_ROOT/delete_gnn.py --dataset $DS --gnn rgcn --random_seed 42 --unlearning_model gnndelete_nodeemb --df in --df_size 2.5 --fullgraph 2>&1 | tail -4 | cut -c1-400
t3=$(date +%s.%N); echo "delete_gnn --fullgraph (${EP1:-100} epochs + eval + test): $(python -c "print(round($t3 - $t2, 1))") s"
python - <<'PY'
import json, glob
f = glob.glob('/tmp/kgrun/checkpoint/synth-biokg/rgcn/gnndelete_nodeemb/*/*/trainer_log.json')[0]
log = json.load(open(f))
h = log['loss_history']
print('loss first/last', h[0][0], h[-1][0], 'dt_auc', log.get('dt_auc'), 'df_auc', log.get('df_auc'))
PY
