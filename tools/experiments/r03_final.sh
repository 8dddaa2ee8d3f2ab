# Everything DESIGN.md quotes for round 3, from one box: the profiled default line (stage table with PMC traffic), the default
# bench line, the R-GCN line, the table of the other configurations.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
TAG=r03_final bash tools/experiments/r03_profile.sh > gpurun_out/r03_final_profile.log 2>&1
cp gpurun_out/r03_final_stages.json profiles/r03_final_stages.json        # (bench.py reads it from profiles/ for in_step_us / traffic)
python bench.py > gpurun_out/r03_bench_default.json 2> gpurun_out/r03_bench_default.err
python bench.py --steps 20 --warmup 5 --no_cached_rate > gpurun_out/r03_bench_driver_shape.json 2>> gpurun_out/r03_bench_default.err
python bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 > gpurun_out/r03_bench_synth_biokg_rgcn.json 2>> gpurun_out/r03_bench_default.err
bash tools/experiments/bench_table.sh > gpurun_out/r03_bench_table.txt 2>&1
tail -c 600 gpurun_out/r03_final_profile.log; cat gpurun_out/r03_bench_table.txt
