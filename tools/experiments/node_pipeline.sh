# BASELINE config 5 end to end on the synthetic collab stand-in: node deletion with GAT (train_node.py -> delete_node.py),
# wall times.  DS / EP0 / EP1 / DF_SIZE override the defaults.
cd $GRAFT_REPO_ROOT
W=/tmp/noderun; rm -rf $W; mkdir -p $W; cd $W
export PYTHONPATH=$GRAFT_REPO_ROOT
DS=${DS:-synth-collab}
ts() { python -c "import time;print(time.time())"; }
el() { python -c "print(f'{$2 - $1:.1f} s')"; }
t0=$(ts)
GNNDELETE_FORCE_EPOCHS=${EP0:-30} GNNDELETE_FORCE_VALID_FREQ=${EP0:-30} timeout 1500 python $GRAFT_REPO_ROOT/train_node.py --dataset $DS --gnn gat --random_seed 42 2>&1 | tail -3 | cut -c1-300
t1=$(ts); echo "train_node gat (${EP0:-30} epochs + eval + test): $(el $t0 $t1)"
GNNDELETE_FORCE_EPOCHS=${EP1:-100} GNNDELETE_FORCE_VALID_FREQ=${EP1:-100} timeout 1500 python $GRAFT_REPO_ROOT/delete_node.py --dataset $DS --gnn gat --random_seed 42 --unlearning_model gnndelete_nodeemb --df in --df_size ${DF_SIZE:-5} 2>&1 | tail -4 | cut -c1-400
t2=$(ts); echo "delete_node gat (${EP1:-100} epochs + eval + test): $(el $t1 $t2)"
