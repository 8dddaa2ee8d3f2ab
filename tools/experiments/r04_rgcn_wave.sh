#!/bin/bash
# wave-private typed conv: parity tests, timing against the tile kernel, the R-GCN bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "rgcn" 2>&1 | tail -5 > gpurun_out/rgcn_wave_tests.txt
cat gpurun_out/rgcn_wave_tests.txt
timeout 900 python tools/experiments/rgcn_wave_time.py 2>&1 | tail -12 | tee gpurun_out/rgcn_wave_time.txt
