# typed conv: diagonal block t on the XCD pair (2 t, 2 t + 1) (GD_RGCN_WAVE_XCD_BLOCKS=1) against the four blocks of a tile on
# one XCD (default): step time of the biokg request and FETCH_SIZE of the three typed launches (separate --pmc passes)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p /tmp/pmc gpurun_out
OUT=gpurun_out/r05_rgcn_xcd_blocks.txt
rm -f $OUT
ARGS="bench.py --workload synth-biokg --gnn rgcn --df in --df_size 2.5 --no_cpu_baseline --no_cached_rate"
for rep in 1 2; do
for mode in 0 1; do
  echo "GD_RGCN_WAVE_XCD_BLOCKS=$mode" >> $OUT
  GD_RGCN_WAVE_XCD_BLOCKS=$mode python $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'final loss', d['final_loss'])" >> $OUT
done; done
for mode in 0 1; do
  rm -rf /tmp/pmc/x$mode
  GD_RGCN_WAVE_XCD_BLOCKS=$mode timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pmc/x$mode -o p -- python $ARGS --steps 12 --warmup 4 --repeats 1 > /tmp/pmc/x$mode.log 2>&1
  echo "== FETCH_SIZE (KiB-units; x2 = bytes / 1024, gfx950) GD_RGCN_WAVE_XCD_BLOCKS=$mode" >> $OUT
  python tools/rocpd_pmc.py /tmp/pmc/x$mode/p_results.db rgcn_wave_kernel >> $OUT 2>&1 || tail -3 /tmp/pmc/x$mode.log >> $OUT
done
cat $OUT
