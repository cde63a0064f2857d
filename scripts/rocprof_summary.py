#!/usr/bin/env python
"""Summarise rocprofv3 sqlite output (rocpd) into small text files for profiles/.

  kernel stats : python scripts/rocprof_summary.py stats <results.db> > profiles/xxx.txt
  pmc counters : python scripts/rocprof_summary.py pmc <results.db> [...] > profiles/xxx.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'^void ', '', name)
    m = re.match(r'(specmi::[a-z0-9_]+(<[^>]*>)?)', name)
    if m:
        return m.group(1)
    return (name[:60] + '...') if len(name) > 60 else name


def stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute('select name,total_calls,total_duration,average,percentage from top_kernels'))
    print(f'# rocprofv3 --kernel-trace --stats   ({db})')
    print(f'{"kernel":58s} {"calls":>7s} {"total_us":>12s} {"avg_us":>10s} {"pct":>6s}')
    for n, calls, tot, avg, pct in rows:
        print(f'{short(n):58s} {calls:7d} {tot / 1e3 if tot > 1e7 else tot:12.1f} {avg:10.3f} {pct:6.2f}')


def pmc(dbs):
    for db in dbs:
        c = sqlite3.connect(db)
        print(f'# rocprofv3 --kernel-trace --pmc   ({db})')
        try:
            cur = c.execute('select * from counters_collection limit 1')
            cols = [d[0] for d in cur.description]
        except Exception as e:
            print('no counters_collection view:', e)
            continue
        namec = 'kernel_name' if 'kernel_name' in cols else 'name'
        q = (f'select {namec}, counter_name, count(*), sum(value), avg(value) from counters_collection '
             f'group by {namec}, counter_name order by {namec}, counter_name')
        last = None
        for kn, cn, n, s, a in c.execute(q):
            k = short(kn)
            if k != last:
                print(k)
                last = k
            print(f'    {cn:32s} dispatches={n:6d} sum={s:18.1f} avg={a:16.1f}')


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
