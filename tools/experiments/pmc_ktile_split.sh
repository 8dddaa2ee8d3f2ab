cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/k$i -o p -- python tools/experiments/ktile_split_probe.py > /tmp/pmc/k$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/k$i/p_results.db "gemm_ktile_split" || tail -3 /tmp/pmc/k$i.log
  python tools/rocpd_pmc.py /tmp/pmc/k$i/p_results.db "gemm_ktile_mfma" || true
done
