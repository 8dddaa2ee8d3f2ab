# counters of the d = 64 forms (VERDICT r5 item 1 asks for VALU instructions per launch before / after + addresser busy)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
out=gpurun_out/r06_spmm_forms_pmc.txt
: > $out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU" \
           "TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc/sf$i
  timeout 400 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/sf$i -o p -- python tools/experiments/r06_spmm_forms.py > /tmp/pmc/sf$i.log 2>&1
  echo "== $set" >> $out
  python tools/rocpd_pmc.py /tmp/pmc/sf$i/p_results.db spmm_persist >> $out 2>&1 || tail -5 /tmp/pmc/sf$i.log >> $out
done
cat $out
