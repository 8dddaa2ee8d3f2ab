cd $GRAFT_REPO_ROOT
W=/tmp/collabrun; rm -rf $W; mkdir -p $W; cd $W
export PYTHONPATH=$GRAFT_REPO_ROOT
python $GRAFT_REPO_ROOT/prepare_dataset.py --dataset synth-collab --seeds 42 > /dev/null 2>&1
GNNDELETE_FORCE_EPOCHS=10 GNNDELETE_FORCE_VALID_FREQ=10 python $GRAFT_REPO_ROOT/train_gnn.py --dataset synth-collab --gnn gcn --random_seed 42 > /dev/null 2>&1
GNNDELETE_FORCE_EPOCHS=200 GNNDELETE_FORCE_VALID_FREQ=200 python -m cProfile -o /tmp/p.prof $GRAFT_REPO_ROOT/delete_gnn.py --dataset synth-collab --gnn gcn --random_seed 42 --unlearning_model gnndelete_nodeemb --df in --df_size 5 > /dev/null 2>&1
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/p.prof'); p.sort_stats('cumulative').print_stats(45)
PY
