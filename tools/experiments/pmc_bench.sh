# Kernel-trace + HBM-traffic counter passes of the default bench line (separate rocprofv3 runs, as the
# MI355X guide prescribes); only text summaries leave the box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
TAG=${TAG:-r01_c}
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline --pretrain_epochs 0 > /tmp/pmc/kt.log 2>&1
python tools/rocpd_summary.py /tmp/pmc/kt/p_results.db gpurun_out/${TAG}_kernel_stats.md > /dev/null
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db loss_finalize 40 > gpurun_out/${TAG}_step_timeline.md 2>&1
grep metric /tmp/pmc/kt.log > gpurun_out/${TAG}_bench_under_rocprof.json
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc/f -o p -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc/w -o p -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/w.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d /tmp/pmc/t -o p -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/t.log 2>&1
(echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline --no_cached_rate ; three separate passes: {FETCH_SIZE}, {WRITE_SIZE}, {TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum}; per-dispatch averages"; for k in f w t; do python tools/rocpd_pmc.py /tmp/pmc/$k/p_results.db gd::; done) > gpurun_out/${TAG}_pmc_per_kernel.txt 2>&1
grep -E "spmm_persist|rows_gemm_mfma|rows_wgrad" gpurun_out/${TAG}_pmc_per_kernel.txt | cut -c1-150
