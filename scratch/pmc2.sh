cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
export SORTED=1 GD_SPMM_PERSIST=1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/pmc/a -o p -- python scratch/spmm_micro.py > /tmp/pmc/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d /tmp/pmc/b -o p -- python scratch/spmm_micro.py > /tmp/pmc/b.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum -d /tmp/pmc/c -o p -- python scratch/spmm_micro.py > /tmp/pmc/c.log 2>&1
for k in a b c; do python tools/rocpd_pmc.py /tmp/pmc/$k/p_results.db spmm_persist; done
tail -2 /tmp/pmc/c.log
