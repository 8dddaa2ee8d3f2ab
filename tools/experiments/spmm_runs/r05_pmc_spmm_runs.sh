# PMC counters of the run-item SpMM against the item kernel on the bench graph (tools/experiments/spmm_runs_micro.py): separate passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
           "TA_BUSY_avr TA_TA_BUSY_sum SQ_INSTS_VMEM_RD" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/r$i -o p -- python tools/experiments/spmm_runs_micro.py > /tmp/pmc/r$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/r$i/p_results.db "spmm_runs_kernel" || tail -3 /tmp/pmc/r$i.log
  python tools/rocpd_pmc.py /tmp/pmc/r$i/p_results.db "spmm_persist_kernel"
done 2>&1 | tee gpurun_out/r05_spmm_runs_pmc.txt
