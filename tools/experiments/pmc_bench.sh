cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline > /tmp/pmc/kt.log 2>&1
python tools/rocpd_summary.py /tmp/pmc/kt/p_results.db gpurun_out/r01_kernel_stats.md > /dev/null
grep metric /tmp/pmc/kt.log > gpurun_out/r01_bench_under_rocprof.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc/f -o p -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline > /tmp/pmc/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pmc/w -o p -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline > /tmp/pmc/w.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum -d /tmp/pmc/t -o p -- python bench.py --steps 10 --warmup 2 --no_cpu_baseline > /tmp/pmc/t.log 2>&1
(for k in f w t; do python tools/rocpd_pmc.py /tmp/pmc/$k/p_results.db gd::; done) > gpurun_out/r01_pmc_per_kernel.txt 2>&1
head -40 gpurun_out/r01_pmc_per_kernel.txt | cut -c1-140
