# GraphSAGE layer 1: the root term through gd_rows_gemm_accumulate_f32 (default) against the aggregation's self-row operand
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "weight_stationary" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_full_size_gpu.py -x -q -k "collab-sage" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -k "sage" 2>&1 | tail -3
rm -f gpurun_out/r06_sage_root_ab.txt
for rep in 1 2 3; do
for v in 1 0; do
  echo "GD_SAGE_ROOT_IN_SPMM=$v" >> gpurun_out/r06_sage_root_ab.txt
  GD_SAGE_ROOT_IN_SPMM=$v timeout 600 python bench.py --gnn sage --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1))" >> gpurun_out/r06_sage_root_ab.txt
done; done
cat gpurun_out/r06_sage_root_ab.txt
