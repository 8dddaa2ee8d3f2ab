# rows_gemm launches with fewer tiles than wave slots: one block per CU, wave-major tile hand-out (GD_ROWS_GEMM_SPREAD=0: before)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or linear or del or mfma or rows" 2>&1 | tail -4 > gpurun_out/r05_spread_test.log
rm -f gpurun_out/r05_spread_ab.txt
run() { python bench.py "$@" --no_cpu_baseline --no_cached_rate --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'final loss', d['final_loss'])" >> gpurun_out/r05_spread_ab.txt; }
for rep in 1 2; do
for mode in 0 1; do
  export GD_ROWS_GEMM_SPREAD=$mode
  echo "GD_ROWS_GEMM_SPREAD=$mode synth-biokg rgcn" >> gpurun_out/r05_spread_ab.txt; run --workload synth-biokg --gnn rgcn --df in --df_size 2.5
  echo "GD_ROWS_GEMM_SPREAD=$mode synth-cora gcn" >> gpurun_out/r05_spread_ab.txt; run --workload synth-cora
  echo "GD_ROWS_GEMM_SPREAD=$mode synth-dblp gcn" >> gpurun_out/r05_spread_ab.txt; run --workload synth-dblp
  echo "GD_ROWS_GEMM_SPREAD=$mode synth-collab gcn (headline)" >> gpurun_out/r05_spread_ab.txt; run
done; done
cat gpurun_out/r05_spread_test.log gpurun_out/r05_spread_ab.txt
