# round 6, third GPU session: the whole GPU suite with durations, the SpMM form counters, the stage tables of dblp / node deletion
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=60 > gpurun_out/r06_c_suite.log 2>&1
tail -75 gpurun_out/r06_c_suite.log
bash tools/experiments/r06_spmm_forms_pmc.sh > /dev/null 2>&1
cat gpurun_out/r06_spmm_forms_pmc.txt | cut -c1-140
WORKLOAD=synth-dblp EXTRA="--df out --df_size 2.5" TAG=r06_c_dblp bash tools/experiments/r06_profile.sh > gpurun_out/r06_c_dblp_profile.log 2>&1
tail -3 gpurun_out/r06_c_dblp_profile.log; cat gpurun_out/r06_c_dblp_step_timeline.md | cut -c1-150
WORKLOAD=synth-collab-nodecls EXTRA="--df_size 5" GNN=gat TAG=r06_c_nodecls_gat bash tools/experiments/r06_profile.sh > gpurun_out/r06_c_nodecls_gat_profile.log 2>&1
tail -3 gpurun_out/r06_c_nodecls_gat_profile.log; cat gpurun_out/r06_c_nodecls_gat_step_timeline.md | cut -c1-150
