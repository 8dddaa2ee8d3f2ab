#!/usr/bin/env python3
"""The per-kernel state table of DESIGN.md section 4, generated from the committed measurements:
  profiles/r06_bench_default.json   (extras.stage_rooflines: algorithmic work, live timing, in-step timing, traffic)
  profiles/r06_final_stages.json    (tools/rocpd_stage_table.py over the rocprofv3 kernel-trace + PMC passes)
-> markdown on stdout, profiles/r06_kernel_table.json (the same rows, machine-checkable).  No GPU needed."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NEXT = {
    'xw1': 'weight-stationary register form (rows_gemm_ws.hip, round 4: 88.5 -> 72 us); at the 1.95-2.2 GHz the part sustains here the matrix pipe alone needs 59 us - left: the tail (14.4 -> 15 units per wave), the launch and drain (a one-unit launch takes 8.8 us; the weight prologue itself is hidden behind the row loads of the first unit - a packed operand image that skips it changes nothing, NOTES round 4)',
    'spmm1': 'at the fabric rate (traffic 628 MB at 6.0-6.4 TB/s); ceiling 0.41 on this graph with per-XCD row ranges (computed from the committed floor / probe records)',
    'del1': 'as xw1 (+ packed sign bits merged with v_permlane swaps): 68.7 -> 62 us',
    'del1_loss_wgrad1': 'round 5: the previous iteration\'s conv2 input gradient (dt2[S1] W2, gated by the stored sign pattern) + Del-1 + folded layer-1 loss + the W_D1 weight gradient in ONE weight-stationary pass (was 59 + 83 + 40 us in three launches, 690 MB; now ~320 MB): instruction-bound (one wave per SIMD issues in order; 320 matrix instructions per 16-row unit and wave at the ~2.0 GHz the part sustains are 5.1 us of its ~6.9); both weights are LDS images (register-resident fragments leave no room for the third product); row-register rings, sched_group_barrier interleaving and a four-way column split all measured equal or slower (NOTES round 5)',
    'wgrad1': 'memory-side (366 MB algorithmic = 61 us at the fabric rate): 2 blocks per CU alternate fetch and MFMA phases; the output-stationary register form measured SLOWER (93 us, NOTES round 4) - it needs more rows in flight per CU, not fewer waves',
    't2': 'weight-stationary form with the two-buffer row selector and ReLU in the operand path: 52.1 -> 45 us; 193 MB at 4.3 TB/s',
    'spmm2': 'latency / window-bound at 4.2-4.8 TB/s of traffic; round 6: eight lanes x two float4 per row (half the visits\' crossbar reads and address products) is 20-32 % SLOWER back to back and +6 % on the step - a row\'s two 128-byte lines asked for by different instructions; the weight stream is worth 3.7 / 1.6 us, all of it the load (profiles/r06_spmm_forms.txt, r06_spmm_forms_pmc.txt)',
    'del2_loss_bwd': 'weight-stationary form (round 4: W_D twice + the 64 x 64 dW sums in registers, 16-row units, no scratch): 62 -> 52 us; the three products of a unit are serial in one wave (loss arithmetic and the LDS transposition between them)',
    'spmm2_t': 'as spmm2 (transposed CSR, S1 rows feed the next product)',
    'dh': 'weight-stationary form with the gate bits fetched a unit ahead: 47.2 -> 41 us; 148 MB at 3.6 TB/s',
    'tail': 'both split-K reductions + Adam + loss finalize in one launch (gd_step_tail_f32); launch-sized',
}
EVID = 'profiles/r06_final_stages.json, r06_final_step_timeline.md'


def main():
    with open(os.path.join(ROOT, 'profiles', 'r06_bench_default.json')) as f:
        line = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(ROOT, 'profiles', 'r06_final_stages.json')) as f:
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
    with open(os.path.join(ROOT, 'profiles', 'r06_kernel_table.json'), 'w') as f:
        json.dump({'step_us_under_rocprof': st['step_span_us'], 'ms_per_step_bench': line['ms_per_step'], 'rows': rows}, f, indent=1)
    print('| stage | kernel | in step us | GF / MB (algorithmic) | bound | fraction in step | PMC traffic (x algorithmic) | state / next |')
    print('|---|---|---|---|---|---|---|---|')
    for r in rows:
        work = f"{r['gflop']} / {r['algorithmic_mb']}" if r['gflop'] is not None else '-'
        frac = (f"{r['frac_in_step']:.2f} ({r['tflops_in_step']} TF)" if r['bound'] == 'mfma' else
                f"{r['frac_in_step']:.2f} ({int(r['gbs_in_step'])} GB/s)") if r['frac_in_step'] is not None else '-'
        tr = f"{r['traffic_mb']} MB ({r['traffic_over_algorithmic']} x)" if r['traffic_over_algorithmic'] else (f"{r['traffic_mb']} MB" if r['traffic_mb'] else '-')
        print(f"| {r['stage']} | `{r['kernel'][:64]}` | {r['in_step_us']} | {work} | {r['bound']} | {frac} | {tr} | {r['next']} |")
    print(f"\nsum of the launches under rocprofv3: {st['sum_in_step_us']:.0f} us; bench (no profiler): {1e3 * line['ms_per_step']:.0f} us per step")


if __name__ == '__main__':
    main()
