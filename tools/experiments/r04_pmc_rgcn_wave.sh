# PMC counters of the wave-private typed conv (layer 1 of the biokg request, tools/experiments/rgcn_wave_time.py): separate passes
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
i=0
for set in "SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" \
           "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD TA_BUSY_avr TA_TA_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/w$i -o p -- python tools/experiments/rgcn_wave_time.py > /tmp/pmc/w$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/w$i/p_results.db "rgcn_wave_kernel<32, 32, 64>" || tail -3 /tmp/pmc/w$i.log
done 2>&1 | tee gpurun_out/r04_rgcn_wave_pmc.txt
