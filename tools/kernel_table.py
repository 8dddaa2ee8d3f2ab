#!/usr/bin/env python3
"""The per-kernel state table of DESIGN.md section 4, generated from the committed measurements:
  profiles/r03_bench_default.json   (extras.stage_rooflines: algorithmic work, live timing, in-step timing, traffic)
  profiles/r03_final_stages.json    (tools/rocpd_stage_table.py over the rocprofv3 kernel-trace + PMC passes)
-> markdown on stdout, profiles/r03_kernel_table.json (the same rows, machine-checkable).  No GPU needed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEXT = {
    'xw1': 'MFMA pipe ~57 % busy at 2.0-2.3 GHz (NOTES: rounds 1-2); tile queue ported (no step-level gain); opt-in split products 69 -> 53 us',
    'spmm1': 'at the fabric rate (traffic 629 MB at 6.1-6.5 TB/s); ceiling 0.41 on this graph with per-XCD row ranges - nothing left but the order',
    'del1': 'as xw1; fusing it with wgrad1 does not fit the LDS (NOTES round 3)',
    'wgrad1': 'between the roofs (0.53 HBM / 0.44 MFMA): 2 blocks per CU alternate fetch and MFMA phases; a third block needs single-buffered tiles',
    't2': 'HBM-side of the ridge at 128 -> 64 (181 MB): 3.5 TB/s',
    'spmm2': 'latency-bound: two light rows per visit (multi-row items) took it 72 -> 67 us; four rows cost a fifth wave slot per SIMD and gave it back (NOTES round 3)',
    'del2_loss_bwd': 'three products on three row streams since the W_D2 weight gradient moved in (was 50 + 26 us in two launches); 256 VGPRs + 56 B/lane of scratch at 2 waves per SIMD',
    'spmm2_t': 'as spmm2 (transposed CSR, S1 rows feed the next product)',
    'dh': 'MFMA / HBM (64 -> 128, gated): 168 MB at 3.6 TB/s',
    'tail': 'both split-K reductions + Adam + loss finalize in one launch (gd_step_tail_f32); launch-sized',
}
EVID = 'profiles/r03_final_stages.json, r03_final_step_timeline.md'


def main():
    with open(os.path.join(ROOT, 'profiles', 'r03_bench_default.json')) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, 'profiles', 'r03_final_stages.json')) as f:
        st = json.load(f)
    roof = {e['stage']: e for e in line['extras']['stage_rooflines']}
    rows = []
    for key, ps in st['stages'].items():
        e = roof.get(key, {})
        ins = e.get('in_step', {})
        rows.append({'stage': key, 'kernel': ps['kernel'], 'in_step_us': round(ps['in_step_us'], 1),
                     'live_us': round(e['avg_us'], 1) if e else None,
                     'gflop': round(e['algorithmic_flops'] / 1e9, 2) if e else None,
                     'algorithmic_mb': round(e['algorithmic_bytes'] / 1e6, 1) if e else None,
                     'bound': e.get('bound', 'latency'),
                     'frac_in_step': round(ins['frac'], 3) if ins else None,
                     'tflops_in_step': round(ins['tflops'], 1) if ins else None,
                     'gbs_in_step': round(ins['gbs'], 0) if ins else None,
                     'traffic_mb': round(ps['traffic_bytes'] / 1e6, 1) if ps.get('traffic_bytes') else None,
                     'traffic_over_algorithmic': round(e['traffic_over_algorithmic'], 2) if e.get('traffic_over_algorithmic') else None,
                     'evidence': EVID, 'next': NEXT.get(key, '')})
    with open(os.path.join(ROOT, 'profiles', 'r03_kernel_table.json'), 'w') as f:
        json.dump({'step_us_under_rocprof': st['step_span_us'], 'ms_per_step_bench': line['ms_per_step'], 'rows': rows}, f, indent=1)
    print('| stage | kernel | in step us | GF / MB (algorithmic) | bound | fraction in step | PMC traffic (x algorithmic) | state / next |')
    print('|---|---|---|---|---|---|---|---|')
    for r in rows:
        work = f"{r['gflop']} / {r['algorithmic_mb']}" if r['gflop'] is not None else '-'
        frac = (f"{r['frac_in_step']:.2f} ({r['tflops_in_step']} TF)" if r['bound'] == 'mfma' else
                f"{r['frac_in_step']:.2f} ({int(r['gbs_in_step'])} GB/s)") if r['frac_in_step'] is not None else '-'
        tr = f"{r['traffic_mb']} MB ({r['traffic_over_algorithmic']} x)" if r['traffic_over_algorithmic'] else (f"{r['traffic_mb']} MB" if r['traffic_mb'] else '-')
        print(f"| {r['stage']} | `{r['kernel'][:58]}` | {r['in_step_us']} | {work} | {r['bound']} | {frac} | {tr} | {r['next']} |")
    print(f"\nsum of the launches under rocprofv3: {st['sum_in_step_us']:.0f} us; bench (no profiler): {1e3 * line['ms_per_step']:.0f} us per step")


if __name__ == '__main__':
    main()
