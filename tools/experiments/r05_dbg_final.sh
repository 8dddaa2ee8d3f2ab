cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
rm -rf /tmp/pmc/kt
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/pmc/kt -o p -- python bench.py --gnn gcn --steps 40 --warmup 10 --repeats 1 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 > /tmp/pmc/kt.log 2>&1
echo "rc=$?"; tail -5 /tmp/pmc/kt.log | cut -c1-300; ls /tmp/pmc/kt | head
echo ---- gat bench
AMD_SERIALIZE_KERNEL=3 HIP_LAUNCH_BLOCKING=1 timeout 300 python bench.py --gnn gat --no_cpu_baseline --no_cached_rate --steps 5 --warmup 2 2>&1 | tail -12 | cut -c1-400
