# SQ counters of the fused aggregate+transform kernel (tools/experiments/spmm_step_micro.py), two passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVES SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/a$i -o p -- python tools/experiments/spmm_step_micro.py > /tmp/pmc/a$i.log 2>&1
  echo "== set $i: $set"
  python tools/rocpd_pmc.py /tmp/pmc/a$i/p_results.db agg_gemm_kernel || tail -5 /tmp/pmc/a$i.log
done
