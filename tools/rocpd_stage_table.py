#!/usr/bin/env python3
"""Per-stage table of the replayed Del step from rocprofv3 rocpd databases (bench.py under --kernel-trace, plus optional
--pmc FETCH_SIZE / WRITE_SIZE passes): the kernels between two consecutive step-end markers, averaged BY POSITION over
all steady-state steps, with the counter traffic of the same kernel name next to them (gfx950 correction of the MI355X
guide: FETCH_SIZE counts 32 B per request where the 16-B/lane gathers move 64 -> doubled; WRITE_SIZE as is; KiB units).

  python tools/rocpd_stage_table.py kt.db [--fetch f.db] [--write w.db] [--marker loss_finalize] [--out stages.json]
        [--stages name,name,...]   (labels by position; default = the GCN both_layerwise step; 'auto' = k00, k01, ... for any step)
"""
import argparse
import json
import sqlite3

GCN_STEP = ['xw1', 'spmm1', 'del1_loss_wgrad1', 't2', 'spmm2', 'del2_loss_bwd', 'spmm2_t', 'tail']          # (chained Del-1 pass: dh is formed in it)
GCN_STEP_R05A = ['xw1', 'spmm1', 'del1_loss_wgrad1', 't2', 'spmm2', 'del2_loss_bwd', 'spmm2_t', 'dh', 'tail']   # (GD_DEL1_CHAIN=0)
GCN_STEP_R04 = ['xw1', 'spmm1', 'del1', 'wgrad1', 't2', 'spmm2', 'del2_loss_bwd', 'spmm2_t', 'dh', 'tail']     # (GD_DEL1_FUSED=0: two launches)


def kernels(db_path):
    db = sqlite3.connect(db_path)
    return db.execute('select name, start, end from kernels order by start').fetchall()


def counter_avg(db_path, counter):
    db = sqlite3.connect(db_path)
    cols = [r[1] for r in db.execute('pragma table_info(counters_collection)')]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    q = (f'select {name_col}, count(*), sum(value) from counters_collection where counter_name = ? group by {name_col}')
    return {n: tot / c for n, c, tot in db.execute(q, (counter,)).fetchall()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('db')
    ap.add_argument('--fetch')
    ap.add_argument('--write')
    ap.add_argument('--marker', default='step_tail')
    ap.add_argument('--stages', default=','.join(GCN_STEP))
    ap.add_argument('--skip', type=int, default=8, help='leading steps left out (warm-up, eager capture passes)')
    ap.add_argument('--out')
    a = ap.parse_args()
    rows = kernels(a.db)
    marks = [i for i, r in enumerate(rows) if a.marker in r[0]]
    steps = [rows[marks[j] + 1: marks[j + 1] + 1] for j in range(len(marks) - 1)]
    labels = a.stages.split(',')
    if a.stages == 'auto':               # any step: the most frequent kernel count between two markers, labels by position
        from collections import Counter
        k = Counter(len(s) for s in steps[a.skip:]).most_common(1)[0][0]
        labels = [f'k{p:02d}' for p in range(k)]
    steady = [s for s in steps[a.skip:] if len(s) == len(labels)]
    assert steady, f'no step with {len(labels)} kernels between markers (lengths seen: {sorted(set(len(s) for s in steps))})'
    fetch = counter_avg(a.fetch, 'FETCH_SIZE') if a.fetch else {}
    write = counter_avg(a.write, 'WRITE_SIZE') if a.write else {}
    out = {'steps_averaged': len(steady), 'kernels_per_step': len(labels), 'stages': {}}
    wall = [(s[-1][2] - steps[a.skip:][0][0][1]) for s in steady]   # unused, kept for debugging
    tot = 0.0
    for p, lab in enumerate(labels):
        durs = [(s[p][2] - s[p][1]) / 1e3 for s in steady]
        name = steady[0][p][0]
        e = {'kernel': name.split('(')[0].replace('void ', '')[:80], 'in_step_us': sum(durs) / len(durs), 'min_us': min(durs),
             'max_us': max(durs)}
        if name in fetch or name in write:
            f_kib, w_kib = fetch.get(name), write.get(name)
            e['fetch_size_kib'] = f_kib
            e['write_size_kib'] = w_kib
            if f_kib is not None and w_kib is not None:
                e['traffic_bytes'] = int(2 * f_kib * 1024 + w_kib * 1024)
        out['stages'][lab] = e
        tot += e['in_step_us']
    out['sum_in_step_us'] = tot
    span = [(s[-1][2] - s[0][1]) / 1e3 for s in steady]
    out['step_span_us'] = sum(span) / len(span)
    txt = json.dumps(out, indent=1)
    if a.out:
        with open(a.out, 'w') as f:
            f.write(txt + '\n')
    print(txt)


if __name__ == '__main__':
    main()
