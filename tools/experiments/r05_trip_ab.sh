# per-trip row test in the two-row items of the 64-float aggregations (add_trip, csrc/spmm.hip): A/B of the bench step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_kernels_gpu.py -x -q -k "spmm or typed_conv or onepass or balanced" 2>&1 | tail -8 > gpurun_out/r05_trip_test.log
rm -f gpurun_out/r05_trip_ab.txt
for rep in 1 2 3; do
for lib in gnndelete_amd/lib/libgd_old_spmm.so ""; do
  echo "lib=${lib:-new}" >> gpurun_out/r05_trip_ab.txt
  GNNDELETE_HIP_LIB=$lib python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'spmm_d64 live us', round(d['extras']['roofline_spmm_d64']['avg_us'],1))" >> gpurun_out/r05_trip_ab.txt
done; done
cat gpurun_out/r05_trip_test.log; cat gpurun_out/r05_trip_ab.txt
