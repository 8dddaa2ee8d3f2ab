# R-GCN step: relu(z1) formed in the operand paths of conv2's two products (GD_RGCN_RELU_PASS=1: the clamp pass): tests + A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -x -q -k "rgcn or gat_backward or kg" 2>&1 | tail -6 > gpurun_out/r05_rgcn_test.log
rm -f gpurun_out/r05_rgcn_relu_ab.txt
for rep in 1 2; do
for mode in 1 0; do
  echo "GD_RGCN_RELU_PASS=$mode" >> gpurun_out/r05_rgcn_relu_ab.txt
  GD_RGCN_RELU_PASS=$mode python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r05_rgcn_relu_ab.txt
done; done
cat gpurun_out/r05_rgcn_test.log; cat gpurun_out/r05_rgcn_relu_ab.txt
