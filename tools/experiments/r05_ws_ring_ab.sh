# two units of rows in flight in the half-width weight-stationary row GEMMs (GD_WS_RING2, csrc/rows_gemm_ws.hip): A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "weight_stationary or rows_gemm or del_loss or dots or rank1" 2>&1 | tail -5 > gpurun_out/r05_ring_test.log
rm -f gpurun_out/r05_ws_ring_ab.txt
for rep in 1 2 3; do
for lib in gnndelete_amd/lib/libgd_ring1.so ""; do
  echo "lib=${lib:-ring2}" >> gpurun_out/r05_ws_ring_ab.txt
  GNNDELETE_HIP_LIB=$lib python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
st=d['extras']['stage_rooflines']
st = st if isinstance(st, list) else list(st.values())
print(round(d['ms_per_step'],4), round(d['value'],1), ' '.join(f\"{e.get('stage')}={e.get('avg_us', e.get('live_us', 0)):.1f}\" for e in st if isinstance(e, dict)))" >> gpurun_out/r05_ws_ring_ab.txt
done; done
cat gpurun_out/r05_ring_test.log; cat gpurun_out/r05_ws_ring_ab.txt
