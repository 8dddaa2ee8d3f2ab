#!/bin/bash
# the overlapped-exchange self-test of bench.py (two ranks over gloo on one GPU), N times: how often do the two programs differ, by how much
cd "$GRAFT_REPO_ROOT" || exit 1
N=${1:-20}
fail=0
for i in $(seq 1 $N); do
  out=$(GD_BENCH_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + i)) bench.py --gpus 2 --probe_partition --probe_overlap --workload synth-small --df in --df_size 5 --gnn ${GNN:-gcn} --loss_type both_layerwise 2>&1)
  if [ $? -ne 0 ]; then fail=$((fail+1)); echo "run $i FAILED: $(echo "$out" | grep 'AssertionError: overlapped' | head -1 | cut -c1-400)"; fi
done
echo "$fail of $N runs differed"
