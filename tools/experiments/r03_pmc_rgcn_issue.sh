# Issue-side counters of the R-GCN typed conv kernels inside the biokg bench step (separate rocprofv3 passes): how many
# instructions of each kind a launch issues and how long the SIMDs spend on them, against the wave-cycles it waits.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
ARGS="bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --steps 3 --warmup 1 --repeats 1 --no_cpu_baseline"
i=0
( for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d /tmp/pmc/q$i -o p -- python $ARGS > /tmp/pmc/q$i.log 2>&1
  echo "== $set"
  python tools/rocpd_pmc.py /tmp/pmc/q$i/p_results.db "rgcn_tile_kernel" | cut -c1-40,70-200 || tail -3 /tmp/pmc/q$i.log
done ) > gpurun_out/r03_rgcn_tile_issue_pmc.txt 2>&1
cat gpurun_out/r03_rgcn_tile_issue_pmc.txt
