# BASELINE config 4 end to end on the synthetic biokg stand-in: prepare -> short original training -> Del
# unlearning (R-GCN, 2.5 % deletion) with few epochs; prints wall times.
cd $GRAFT_REPO_ROOT
W=/tmp/kgrun; rm -rf $W; mkdir -p $W; cd $W
export PYTHONPATH=$GRAFT_REPO_ROOT
DS=${DS:-synth-biokg}
t0=$(date +%s.%N)
python $GRAFT_REPO_ROOT/prepare_dataset.py --dataset $DS --seeds 42 2>&1 | tail -2
t1=$(date +%s.%N); echo "prepare: $(python -c "print(round($t1 - $t0, 1))") s"
GNNDELETE_FORCE_EPOCHS=2 GNNDELETE_FORCE_VALID_FREQ=2 timeout 1500 python $GRAFT_REPO_ROOT/train_gnn.py --dataset $DS --gnn rgcn --random_seed 42 --epochs 2 --valid_freq 2 2>&1 | tail -4
t2=$(date +%s.%N); echo "train_gnn (2 epochs + eval + test): $(python -c "print(round($t2 - $t1, 1))") s"
GNNDELETE_FORCE_EPOCHS=2 GNNDELETE_FORCE_VALID_FREQ=2 timeout 1500 python $GRAFT_REPO_ROOT/delete_gnn.py --dataset $DS --gnn rgcn --random_seed 42 --unlearning_model gnndelete_nodeemb --df in --df_size 2.5 --epochs 2 --valid_freq 2 2>&1 | tail -6
t3=$(date +%s.%N); echo "delete_gnn (2 epochs + eval + test): $(python -c "print(round($t3 - $t2, 1))") s"
