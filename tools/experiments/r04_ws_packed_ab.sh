# weight-stationary row GEMMs with the packed operand image (constant weights) against the LDS prologue: in-step kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for p in 1 0; do
  GD_ROWS_GEMM_WS_PACKED=$p rocprofv3 --kernel-trace --stats -d /tmp/wsab$p -o p -- python bench.py --steps 40 --warmup 5 --repeats 1 --no_cpu_baseline --no_cached_rate > /tmp/wsab$p.log 2>&1
  python tools/rocpd_summary.py /tmp/wsab$p/p_results.db /tmp/wsab$p.md > /dev/null
  echo "== GD_ROWS_GEMM_WS_PACKED=$p"; grep -o '"value": [0-9.]*' /tmp/wsab$p.log | head -1
  grep "rows_gemm_ws\|del_loss_bwd_ws\|rows_wgrad\|spmm_persist\|step_tail" /tmp/wsab$p.md | sed 's/(float const.*`//;s/(int4 const.*`//;s/(HIP_vector.*`//' | cut -c1-140
done 2>&1 | tee gpurun_out/r04_ws_packed_ab.txt
