cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'final loss', d['final_loss'])"; }
for rep in 1 2; do
echo two-launch; GD_DEL1_FUSED=0 run
echo fused-interleaved; run
echo fused-no-interleave; GNNDELETE_HIP_LIB=$GRAFT_REPO_ROOT/tools/experiments/lab/libgd_ni.so run
done
