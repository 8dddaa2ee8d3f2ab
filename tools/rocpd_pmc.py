#!/usr/bin/env python3
"""Per-kernel PMC counter totals (averaged per dispatch) from a rocprofv3 rocpd database.
Usage: python tools/rocpd_pmc.py results.db [kernel-name-substring]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else ''
    cols = [r[1] for r in db.execute('pragma table_info(counters_collection)')]
    name_col = 'kernel_name' if 'kernel_name' in cols else 'name'
    q = (f"select {name_col}, counter_name, count(*), sum(value) from counters_collection "
         f"where {name_col} like ? group by {name_col}, counter_name")
    rows = db.execute(q, (f'%{pat}%',)).fetchall()
    for name, counter, calls, total in rows:
        print(f'{name[:60]:60s} {counter:24s} dispatches={calls:5d} avg={total / calls:16.1f}')


if __name__ == '__main__':
    main()
