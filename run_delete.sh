#!/bin/bash
# Positional launcher with the reference's argument order (run_delete.sh:5-27):
#   ./run_delete.sh DATA MODEL UNLEARNING_MODEL DF DF_SIZE SEED
# e.g. ./run_delete.sh synth-dblp gcn gnndelete_nodeemb out 2.5 42
# (no conda env to activate here; wandb is optional and stays offline)
set -e
if [ "$#" -lt 6 ]; then
  echo "usage: $0 DATA MODEL UNLEARNING_MODEL DF DF_SIZE SEED" >&2
  exit 2
fi
DATA=$1; MODEL=$2; UN=$3; DF=$4; DF_SIZE=$5; SEED=$6
HERE="$(cd "$(dirname "$0")" && pwd)"
export WANDB_MODE=offline
export WANDB_NAME="${UN}_${DATA}_${MODEL}_${DF}_${DF_SIZE}_${SEED}"
export WANDB_RUN_ID="$WANDB_NAME"
exec python "$HERE/delete_gnn.py" --lr 1e-3 --epochs 1500 --dataset "$DATA" --random_seed "$SEED" \
     --unlearning_model "$UN" --gnn "$MODEL" --df "$DF" --df_size "$DF_SIZE"
