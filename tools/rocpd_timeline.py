#!/usr/bin/env python3
"""Timeline of one steady-state step from a rocprofv3 rocpd database: the kernels between two
consecutive launches of a marker kernel (default: the loss finalize kernel that ends a step), with
start offset, duration, queue and the idle gap on the whole device before each start.
Usage: python tools/rocpd_timeline.py x_results.db [marker-substring] [which-step-from-the-end]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else 'loss_finalize'
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols else None)
    q = f"select name, start, end, {qcol or '0'} from kernels order by start"
    rows = db.execute(q).fetchall()
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(marks) < back + 1:
        print('not enough marker launches', len(marks))
        return
    lo, hi = marks[-back - 1] + 1, marks[-back] + 1
    t0 = rows[lo][1]
    busy_end = rows[lo - 1][2]
    print(f'# step = {(rows[hi - 1][2] - rows[lo - 1][2]) / 1e3:.1f} us (end of previous marker to end of this one)')
    print('| start us | dur us | gap us | queue | kernel |')
    print('|---|---|---|---|---|')
    tot = 0
    for name, s, e, qid in rows[lo:hi]:
        gap = max(0, s - busy_end)
        busy_end = max(busy_end, e)
        tot += e - s
        print(f'| {(s - t0) / 1e3:8.1f} | {(e - s) / 1e3:7.1f} | {gap / 1e3:5.1f} | {qid} | `{name[:70]}` |')
    print(f'# sum of kernel durations {tot / 1e3:.1f} us')


if __name__ == '__main__':
    main()
