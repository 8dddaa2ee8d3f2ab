#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, share) from a rocprofv3 rocpd SQLite database.
Usage: python tools/rocpd_summary.py gpurun_out/prof/x_results.db [out.md]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ['| kernel | calls | total ms | avg us | min us | max us | % |', '|---|---|---|---|---|---|---|']
    for name, calls, tot, avg, mn, mx in rows:
        lines.append(f'| `{name[:110]}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | '
                     f'{100 * tot / total:.1f} |')
    text = '\n'.join(lines)
    print(text)
    if len(sys.argv) > 2:
        with open(sys.argv[2], 'w') as f:
            f.write(text + '\n')


if __name__ == '__main__':
    main()
