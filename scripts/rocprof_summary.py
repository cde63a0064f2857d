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


def traffic(dbs):
    """HBM traffic per kernel family from FETCH_SIZE / WRITE_SIZE passes (KB units).
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request of a
    wide coalesced stream, so the read side is doubled; WRITE_SIZE is taken as reported.
    Top-level keys describe the dominant family (conv_igemm_f32, what bench.py reports as
    roofline.traffic); 'families' holds the same numbers for the other conv kernels."""
    import json

    def family(pattern):
        tot = {}
        for db in dbs:
            c = sqlite3.connect(db)
            cur = c.execute('select * from counters_collection limit 1')
            cols = [d[0] for d in cur.description]
            namec = 'kernel_name' if 'kernel_name' in cols else 'name'
            q = (f"select counter_name, count(*), sum(value) from counters_collection where {namec} like '%{pattern}%' "
                 f"and counter_name in ('FETCH_SIZE','WRITE_SIZE') group by counter_name")
            for cn, n, sm in c.execute(q):
                tot[cn] = (n, sm)
        if 'FETCH_SIZE' not in tot or 'WRITE_SIZE' not in tot:
            return None
        nf, f = tot['FETCH_SIZE']; nw, w = tot['WRITE_SIZE']
        return {'launches_fetch_pass': nf, 'launches_write_pass': nw, 'FETCH_SIZE_KB_sum': f, 'WRITE_SIZE_KB_sum': w,
                'read_bytes_per_launch': 2.0 * f * 1024 / nf, 'write_bytes_per_launch': w * 1024 / nw,
                'traffic_bytes_per_launch': 2.0 * f * 1024 / nf + w * 1024 / nw}

    out = {'kernel': 'conv_igemm_f32 (all tile variants)'}
    main = family('conv_igemm')
    if main:
        out.update(main)
        out['correction'] = 'read = 2 x FETCH_SIZE (gfx950 counts 64 B per 128-B request), write = WRITE_SIZE'
    fams = {}
    for name, pat in (('conv_igemm_f32<128x128>', 'conv_igemm_f32_kernel<128, 128'), ('conv_igemm_f32<64x64>', 'conv_igemm_f32_kernel<64, 64'),
                      ('conv_wino_f32', 'conv_wino'), ('stem_conv7x7', 'stem_conv7x7'), ('maxpool3x3s2', 'maxpool'),
                      ('smpl_skin', 'smpl_skin')):
        r = family(pat)
        if r:
            fams[name] = r
    out['families'] = fams
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2])
    elif sys.argv[1] == 'traffic':
        traffic(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
