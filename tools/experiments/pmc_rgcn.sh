# PMC counters of the R-GCN typed conv kernels inside the biokg bench step: separate rocprofv3 passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/r$i -o p -- python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --steps 3 --warmup 1 --no_cpu_baseline > /tmp/pmc/r$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/r$i/p_results.db "rgcn_tile_kernel<128, 32, 32>" || tail -3 /tmp/pmc/r$i.log
done
