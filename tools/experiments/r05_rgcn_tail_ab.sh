# one tail launch (both reductions + Adam + finalize) also where the loss / weight-gradient stages run unfused (the R-GCN request): tests + A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests/test_engine_gpu.py tests/test_cli_gpu.py -x -q --durations=8 2>&1 | tail -16 > gpurun_out/r05_tail_test.log
python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline > gpurun_out/r05_rgcn_extras.json 2> gpurun_out/r05_rgcn_extras.err
cat gpurun_out/r05_tail_test.log; python -c "
import json
d=json.loads([l for l in open('gpurun_out/r05_rgcn_extras.json') if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), d.get('extras'))"; tail -3 gpurun_out/r05_rgcn_extras.err
