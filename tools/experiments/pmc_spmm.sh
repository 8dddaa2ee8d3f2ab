# SQ counters of the balanced SpMM on the bench graph (tools/experiments/spmm_micro.py), two passes.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
export CAPS=${CAPS:-8192}
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/s$i -o p -- python tools/experiments/spmm_micro.py > /tmp/pmc/s$i.log 2>&1
  echo "== set $i: $set"
  python tools/rocpd_pmc.py /tmp/pmc/s$i/p_results.db spmm_persist || tail -5 /tmp/pmc/s$i.log
done
