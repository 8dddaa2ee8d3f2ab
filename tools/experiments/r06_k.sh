cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -X faulthandler -m pytest tests/test_kernels_gpu.py -x -v -k "weight_stationary or gram or rbf or loss_zoo or spmm or gat_hub" > gpurun_out/r06_k.log 2>&1
grep -n "PASSED\|FAILED\|ERROR\|Fatal\|Abort\|fault\|HSA\|Memory" gpurun_out/r06_k.log | tail -30
grep -n "Fatal Python error" -A 25 gpurun_out/r06_k.log | head -60
