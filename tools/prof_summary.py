#!/usr/bin/env python
"""Summarise a rocprofv3 `--kernel-trace --stats` run (rocpd sqlite output) as a per-kernel table.

    python tools/prof_summary.py <results.db> [steps] > profiles/<name>.txt
"""
import sqlite3
import sys


def main(path, steps=None):
    con = sqlite3.connect(path)
    rows = con.execute(
        'select name, count(*), sum(end-start)/1000.0, avg(end-start)/1000.0, min(end-start)/1000.0, '
        'max(end-start)/1000.0, max(vgpr_count), max(lds_size) from kernels group by name order by 3 desc').fetchall()
    total = sum(r[2] for r in rows)
    print('# rocprofv3 --kernel-trace --stats summary of %s' % path)
    print('# total kernel time %.1f us over %d dispatches' % (total, sum(r[1] for r in rows)))
    print('%-74s %7s %12s %10s %10s %10s %6s %5s %7s' % ('kernel', 'calls', 'total_us', 'avg_us', 'min_us', 'max_us', 'pct',
                                                          'vgpr', 'lds'))
    for name, n, tot, avg, mn, mx, vgpr, lds in rows:
        print('%-74s %7d %12.1f %10.2f %10.2f %10.2f %6.1f %5s %7s' % (name[:74], n, tot, avg, mn, mx, 100 * tot / total,
                                                                      vgpr, lds))
    if steps:
        print('# per step (%d steps incl. warm-up + model fit pass): %.1f us of kernel time' % (steps, total / steps))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else None)
