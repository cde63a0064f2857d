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


def timeline(db, last=70):
    """The last `last` kernel dispatches in start order: duration and the idle gap since the previous kernel ended
    (a hipGraph replay of a small-batch step: what the kernels take and what the launches cost in between)."""
    c = sqlite3.connect(db)
    rows = None
    for q in ('select name, start, end from kernels order by start',
              'select kernel_name, start, end from kernels order by start',
              'select name, start_timestamp, end_timestamp from kernels order by start_timestamp'):
        try:
            rows = list(c.execute(q))
            break
        except Exception:
            continue
    if rows is None:
        print('no usable kernels view; tables:', [r[0] for r in c.execute("select name from sqlite_master")])
        return
    rows = rows[-last:]
    print(f'# rocprofv3 --kernel-trace: last {len(rows)} dispatches ({db})')
    print(f'{"kernel":58s} {"dur_us":>9s} {"gap_us":>9s}')
    prev_end, tk, tg = None, 0.0, 0.0
    for n, st, en in rows:
        gap = 0.0 if prev_end is None else (st - prev_end) / 1e3
        print(f'{short(n):58s} {(en - st) / 1e3:9.2f} {gap:9.2f}')
        tk += (en - st) / 1e3
        tg += max(gap, 0.0)
        prev_end = en
    print(f'# kernels {tk:.1f} us, gaps {tg:.1f} us, span {(rows[-1][2] - rows[0][1]) / 1e3:.1f} us')


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


def layer_traffic(fetch_db, write_db, layers_json):
    """Per-LAYER traffic: the conv dispatches of the FETCH_SIZE / WRITE_SIZE passes in dispatch order (eager, one stream: the
    same order in every step) against the per-launch table bench.py wrote (label, algorithmic bytes), averaged over the steps."""
    import json
    entries = [e for e in json.load(open(layers_json))['entries'] if e['kernel'].startswith(('conv_', 'stem_', 'maxpool'))
               and 'splitK' not in e['kernel']]

    def series(db, counter):
        c = sqlite3.connect(db)
        cur = c.execute('select * from counters_collection limit 1')
        cols = [d[0] for d in cur.description]
        namec = 'kernel_name' if 'kernel_name' in cols else 'name'
        order = next((k for k in ('dispatch_id', 'start', 'start_timestamp', 'timestamp', 'id') if k in cols), None)
        if order is None:
            raise SystemExit(f'no ordering column in counters_collection: {cols}')
        q = (f"select {namec}, sum(value), min({order}) from counters_collection where counter_name = '{counter}' "
             f"group by {order}, {namec} order by min({order})")
        out = []
        for kn, v, _ in c.execute(q):
            k = short(kn)
            if 'conv_igemm' in k or 'conv_wino' in k or 'stem_conv' in k or 'maxpool' in k:
                out.append((k, v))
        return out

    f, w = series(fetch_db, 'FETCH_SIZE'), series(write_db, 'WRITE_SIZE')
    # split-K dispatches (FC heads) carry the SPLITK template flag: drop them by name
    issplit = lambda k: bool(re.search(r'conv_igemm_f32_kernel<64, 64, 2, 2, true, 32, false, true, true>', k))
    f = [x for x in f if not issplit(x[0])]
    w = [x for x in w if not issplit(x[0])]
    n = len(entries)
    if len(f) % n or len(w) % n or not f:
        raise SystemExit(f'{len(f)} / {len(w)} conv dispatches do not divide into steps of {n} launches')
    steps = len(f) // n
    print(f'# per-layer traffic (MB per launch, average of {steps} steps; read = 2 x FETCH_SIZE, write = WRITE_SIZE) vs algorithmic bytes')
    print(f'{"model":9s} {"layer":34s} {"kernel":44s} {"read":>8s} {"write":>8s} {"sum":>8s} {"algorithmic":>12s} {"ratio":>6s}')
    agg = {}
    for i, e in enumerate(entries):
        rd = sum(2.0 * f[s * n + i][1] * 1024 for s in range(steps)) / steps
        wr = sum(w[s * n + i][1] * 1024 for s in range(len(w) // n)) / (len(w) // n)
        alg = e['bytes']
        print(f'{e["model"]:9s} {e["label"]:34s} {e["kernel"][:44]:44s} {rd / 1e6:8.1f} {wr / 1e6:8.1f} {(rd + wr) / 1e6:8.1f} {alg / 1e6:12.1f} '
              f'{(rd + wr) / alg:6.2f}   [{f[i][0][:60]}]')
        key = re.sub(r'^backbone\.', '', e['label'])
        key = re.sub(r'\.\d+\.', '.', key)
        a = agg.setdefault(key, [0.0, 0.0, 0])
        a[0] += rd + wr; a[1] += alg; a[2] += 1
    print('# by layer type (both networks):')
    for k, (t, a, c_) in sorted(agg.items(), key=lambda kv: -(kv[1][0] - kv[1][1])):
        print(f'  {k:28s} x{c_:<3d} measured {t / 1e6:9.1f} MB  algorithmic {a / 1e6:9.1f} MB  ratio {t / a:5.2f}  excess {(t - a) / 1e6:8.1f} MB')


if __name__ == '__main__':
    if sys.argv[1] == 'layer_traffic':
        layer_traffic(sys.argv[2], sys.argv[3], sys.argv[4])
    elif sys.argv[1] == 'timeline':
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 70)
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2])
    elif sys.argv[1] == 'traffic':
        traffic(sys.argv[2:])
    else:
        pmc(sys.argv[2:])
