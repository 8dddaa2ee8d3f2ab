# lab timing of the d = 64 forms + in-step A/B of the weighted 8x2u2 form (alternating runs of the bench step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python tools/experiments/r06_spmm_forms.py > gpurun_out/r06_spmm_forms.txt 2>&1
tail -30 gpurun_out/r06_spmm_forms.txt
rm -f gpurun_out/r06_spmm_forms_step_ab.txt
for rep in 1 2 3; do
for form in "" 8x2u2; do
  echo "GD_SPMM_D64_FORM=${form:-product}" >> gpurun_out/r06_spmm_forms_step_ab.txt
  GD_SPMM_D64_FORM=$form timeout 600 python bench.py --steps 200 --warmup 20 --no_cpu_baseline --no_cached_rate --pretrain_epochs 0 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print(round(d['ms_per_step'],4), round(d['value'],1), 'spmm_d64 live us', round(d['extras']['roofline_spmm_d64']['avg_us'],1))" >> gpurun_out/r06_spmm_forms_step_ab.txt
done; done
cat gpurun_out/r06_spmm_forms_step_ab.txt
