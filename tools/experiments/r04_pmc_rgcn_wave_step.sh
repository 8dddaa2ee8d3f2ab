# Counters of the wave-private typed conv kernels inside the biokg bench step (separate rocprofv3 passes): memory side
# (what the fabric / L2 carry per launch), issue side, and the per-kernel stats of the same command.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
ARGS="bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --steps 3 --warmup 1 --repeats 1 --no_cpu_baseline"
i=0
( for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum" \
    "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
    "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
    "TA_BUSY_avr TA_TA_BUSY_sum SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/m$i -o p -- python $ARGS > /tmp/pmc/m$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/m$i/p_results.db "rgcn_wave_kernel" | sed 's/(int const.*) *//' | cut -c1-200 || tail -3 /tmp/pmc/m$i.log
done
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pmc/mk -o p -- python $ARGS > /tmp/pmc/mk.log 2>&1
python tools/rocpd_summary.py /tmp/pmc/mk/p_results.db /tmp/pmc/mk.md > /dev/null; cp /tmp/pmc/mk.md gpurun_out/r04_rgcn_kernel_stats.md
grep "rgcn_\|rows_gemm\|gate_rows\|del_" /tmp/pmc/mk.md | cut -c1-90,110-220
grep -o '"value": [0-9.]*' /tmp/pmc/mk.log | head -1 ) > gpurun_out/r04_rgcn_wave_step_pmc.txt 2>&1
cat gpurun_out/r04_rgcn_wave_step_pmc.txt
