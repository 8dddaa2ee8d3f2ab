# Issue-side counters of every kernel of the default bench step (separate rocprofv3 passes): matrix-pipe busy cycles, the share
# of wave-cycles spent waiting / ready-not-issued / executing, instruction mix.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
ARGS="bench.py --steps 40 --warmup 8 --repeats 1 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0"
i=0
( for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/s$i -o p -- python $ARGS > /tmp/pmc/s$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/s$i/p_results.db "gd::" | cut -c1-64,70-200 || tail -3 /tmp/pmc/s$i.log
done ) > gpurun_out/r06_step_issue_pmc.txt 2>&1
cat gpurun_out/r06_step_issue_pmc.txt
