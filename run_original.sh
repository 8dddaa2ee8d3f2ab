#!/bin/bash
# Positional launcher with the reference's argument order (run_original.sh:5-26):
#   ./run_original.sh DATA MODEL SEED        e.g. ./run_original.sh synth-dblp gcn 42
set -e
if [ "$#" -lt 3 ]; then
  echo "usage: $0 DATA MODEL SEED" >&2
  exit 2
fi
DATA=$1; MODEL=$2; SEED=$3
HERE="$(cd "$(dirname "$0")" && pwd)"
export WANDB_MODE=offline
export WANDB_NAME="original_${DATA}_${MODEL}_${SEED}"
export WANDB_RUN_ID="$WANDB_NAME"
exec python "$HERE/train_gnn.py" --lr 1e-3 --epochs 1500 --dataset "$DATA" --random_seed "$SEED" \
     --unlearning_model original --gnn "$MODEL"
