# PMC counters of the split-bf16 lab kernels: separate rocprofv3 passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
./tools/experiments/split_lab.bin 178921 30 256
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/s$i -o p -- ./tools/experiments/split_lab.bin 178921 5 256 > /tmp/pmc/s$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/s$i/p_results.db "split_gemm_kernel<6, 0>" || tail -3 /tmp/pmc/s$i.log
  python tools/rocpd_pmc.py /tmp/pmc/s$i/p_results.db "f32_gemm" || true
done
