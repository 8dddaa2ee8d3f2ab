# GATConv backward without the transposition pass (gd_spmm_csr_onepass_aux_f32): tests + A/B of the GAT bench step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_models_gpu.py -x -q -k "gat or typed_conv or rgat or rgcn" 2>&1 | tail -6 > gpurun_out/r05_gat_test.log
rm -f gpurun_out/r05_gat_aux_ab.txt
for rep in 1 2 3; do
for mode in 1 0; do
  echo "GD_GAT_TRANSPOSE_PASS=$mode" >> gpurun_out/r05_gat_aux_ab.txt
  GD_GAT_TRANSPOSE_PASS=$mode python bench.py --gnn gat --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r05_gat_aux_ab.txt
done; done
cat gpurun_out/r05_gat_test.log; cat gpurun_out/r05_gat_aux_ab.txt
