#!/usr/bin/env python3
"""Per (kernel, grid size) durations from a rocprofv3 rocpd database, in first-launch order.
Usage: python tools/rocpd_bygrid.py x_results.db [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ''
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
gx = 'grid_x' if 'grid_x' in cols else ('grid_size_x' if 'grid_size_x' in cols else None)
q = (f"select name, {gx}, count(*), avg(end-start), min(end-start), min(start) from kernels group by name, {gx} order by 6" if gx else
     "select name, 0, count(*), avg(end-start), min(end-start), min(start) from kernels group by name order by 6")
for name, g, n, avg, mn, _ in db.execute(q):
    if pat in name:
        print(f'{name[:70]:70s} grid {g:>9} calls {n:4d} avg {avg / 1e3:8.1f} us min {mn / 1e3:8.1f} us')
