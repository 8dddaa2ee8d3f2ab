# BASELINE config 3 end to end on the synthetic collab stand-in (GCN and GraphSAGE): prepare -> short original
# training -> Del unlearning 5 % IN (full-graph fused step) -> test; wall times via python.
cd $GRAFT_REPO_ROOT
W=/tmp/collabrun; rm -rf $W; mkdir -p $W; cd $W
export PYTHONPATH=$GRAFT_REPO_ROOT
DS=${DS:-synth-collab}
ts() { python -c "import time;print(time.time())"; }
el() { python -c "print(f'{$2 - $1:.1f} s')"; }
t0=$(ts); python $GRAFT_REPO_ROOT/prepare_dataset.py --dataset $DS --seeds 42 2>&1 | tail -1 | cut -c1-200; t1=$(ts); echo "prepare: $(el $t0 $t1)"
for G in ${GNNS:-gcn sage}; do
GNNDELETE_FORCE_EPOCHS=${EP0:-30} GNNDELETE_FORCE_VALID_FREQ=${EP0:-30} timeout 1500 python $GRAFT_REPO_ROOT/train_gnn.py --dataset $DS --gnn $G --random_seed 42 2>&1 | tail -3 | cut -c1-300
t2=$(ts); echo "train_gnn $G: $(el $t1 $t2)"
GNNDELETE_FORCE_EPOCHS=${EP1:-200} GNNDELETE_FORCE_VALID_FREQ=${EP1:-200} timeout 1500 python $GRAFT_REPO_ROOT/delete_gnn.py --dataset $DS --gnn $G --random_seed 42 --unlearning_model gnndelete_nodeemb --df in --df_size 5 $EXTRA 2>&1 | tail -4 | cut -c1-400
t1=$(ts); echo "delete_gnn $G: $(el $t2 $t1)"
done
