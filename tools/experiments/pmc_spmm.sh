cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc
for v in wave group; do
  if [ $v = group ]; then export GD_SPMM_GROUP=1; else unset GD_SPMM_GROUP; fi
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD -d /tmp/pmc/${v}_sq -o p -- python tools/experiments/spmm_micro.py > /tmp/pmc/${v}_sq.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d /tmp/pmc/${v}_tcc -o p -- python tools/experiments/spmm_micro.py > /tmp/pmc/${v}_tcc.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d /tmp/pmc/${v}_sq2 -o p -- python tools/experiments/spmm_micro.py > /tmp/pmc/${v}_sq2.log 2>&1
  for k in sq tcc sq2; do echo "== $v $k"; python tools/rocpd_pmc.py /tmp/pmc/${v}_${k}/p_results.db spmm_ ; done
done
