# rocprofv3 kernel statistics of the bench on the bag-of-words configurations (is any rocBLAS Cijk_* kernel left?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/prof gpurun_out
for wl in synth-dblp synth-cora; do
  df=out; size=2.5; [ $wl = synth-cora ] && size=0.5
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof/$wl -o p -- python bench.py --workload $wl --df $df --df_size $size --steps 50 --warmup 5 --no_cpu_baseline --no_cached_rate > /tmp/prof/$wl.log 2>&1
  python tools/rocpd_summary.py /tmp/prof/$wl/p_results.db gpurun_out/${TAG:-r02_b}_${wl}_kernel_stats.md > /dev/null
  grep metric /tmp/prof/$wl.log > gpurun_out/${TAG:-r02_b}_${wl}_bench_under_rocprof.json
  echo "$wl: Cijk kernels: $(grep -c Cijk gpurun_out/${TAG:-r02_b}_${wl}_kernel_stats.md)"
  head -12 gpurun_out/${TAG:-r02_b}_${wl}_kernel_stats.md | cut -c1-160
done
