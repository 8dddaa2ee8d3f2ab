cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
ARGS="bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --steps 40 --warmup 5 --repeats 1"
( echo "## GD_RGCN_REORDER=1 (degree-weighted label propagation)"; python $ARGS --cpu_baseline_iters 2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step','parity')}); print(d['roofline'])"
  echo "## without (default)"; GD_RGCN_REORDER=0 python $ARGS --no_cpu_baseline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:d[k] for k in ('value','ms_per_step')}); print(d['roofline'])"
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc/rr -o p -- python $ARGS --steps 3 --warmup 1 --no_cpu_baseline > /tmp/pmc/rr.log 2>&1
  echo "== FETCH_SIZE with the locality order"; python tools/rocpd_pmc.py /tmp/pmc/rr/p_results.db "rgcn_tile_kernel" | cut -c1-40,70-200
  timeout 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc/rh -o p -- python $ARGS --steps 3 --warmup 1 --no_cpu_baseline > /tmp/pmc/rh.log 2>&1
  python tools/rocpd_pmc.py /tmp/pmc/rh/p_results.db "rgcn_tile_kernel" | cut -c1-40,70-200
) > gpurun_out/r03_rgcn_reorder_ab.txt 2>&1
cat gpurun_out/r03_rgcn_reorder_ab.txt
