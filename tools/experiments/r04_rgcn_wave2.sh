#!/bin/bash
# wave-private typed conv in the engine: R-GCN tests, full-size parity, the config-4 bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py tests/test_models_gpu.py tests/test_full_size_gpu.py -q -x -k "rgcn or kg or typed" 2>&1 | tail -5 | tee gpurun_out/rgcn_wave_engine_tests.txt
timeout 900 python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r04_bench_synth_biokg_rgcn.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_synth_biokg_rgcn.json').read())
print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600])
PY
