# kernel-trace profile of the default bench line; writes only text summaries to gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --steps 30 --warmup 5 --no_cpu_baseline "$@" > /tmp/pmc/kt.log 2>&1
python tools/rocpd_summary.py /tmp/pmc/kt/p_results.db gpurun_out/kernel_stats.md > /dev/null
grep metric /tmp/pmc/kt.log > gpurun_out/bench_under_rocprof.json
head -30 gpurun_out/kernel_stats.md | cut -c1-200
python tools/rocpd_timeline.py /tmp/pmc/kt/p_results.db loss_finalize 40 > gpurun_out/timeline.md 2>&1
cat gpurun_out/timeline.md | cut -c1-160
