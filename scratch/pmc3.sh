cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
python scratch/gemm_micro.py 2>&1 | tail -2
DOUT=64 python scratch/gemm_micro.py 2>&1 | tail -2
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d /tmp/pmc/g1 -o p -- python scratch/gemm_micro.py > /tmp/pmc/g1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU -d /tmp/pmc/g2 -o p -- python scratch/gemm_micro.py > /tmp/pmc/g2.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum -d /tmp/pmc/g3 -o p -- python scratch/gemm_micro.py > /tmp/pmc/g3.log 2>&1
for k in g1 g2 g3; do python tools/rocpd_pmc.py /tmp/pmc/$k/p_results.db rows_gemm; done
