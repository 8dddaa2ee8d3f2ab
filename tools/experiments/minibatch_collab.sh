cd $GRAFT_REPO_ROOT
W=/tmp/collabrun; rm -rf $W; mkdir -p $W; cd $W
export PYTHONPATH=$GRAFT_REPO_ROOT
python $GRAFT_REPO_ROOT/prepare_dataset.py --dataset synth-collab --seeds 42 > /dev/null 2>&1
GNNDELETE_FORCE_EPOCHS=10 GNNDELETE_FORCE_VALID_FREQ=10 python $GRAFT_REPO_ROOT/train_gnn.py --dataset synth-collab --gnn gcn --random_seed 42 > /dev/null 2>&1
python - <<'PY'
import time; t=time.time()
import subprocess, os, sys
env = dict(os.environ, GNNDELETE_FORCE_EPOCHS='2', GNNDELETE_FORCE_VALID_FREQ='2')
r = subprocess.run([sys.executable, os.environ['GRAFT_REPO_ROOT'] + '/delete_gnn.py', '--dataset', 'synth-collab', '--gnn', 'gcn', '--random_seed', '42', '--unlearning_model', 'gnndelete_nodeemb', '--df', 'in', '--df_size', '5', '--minibatch'], env=env, capture_output=True, text=True)
print(r.stdout[-1500:]); print(r.stderr[-1500:]); print('wall', round(time.time() - t, 1), 's rc', r.returncode)
PY
